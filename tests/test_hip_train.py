"""GPU tests of the pieces around the rasteriser: fused pre-activation path, event loss, Adam, distCUDA2,
and the fused training step against the autograd (reference-style) step."""
import math
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from helpers import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scene(N=3000, W=176, H=128, seed=0):
    from event_3dgs_amd import synth
    from event_3dgs_amd.cameras import orbit_camera
    params = synth.make_scene(N, "trained", seed=seed, device=DEV)
    cams = [orbit_camera(0, 16, W, H, device=DEV, daz=d) for d in (0.0, 0.004, 0.012)]
    return params, cams


def _gts(params, cams, bg):
    from event_3dgs_amd.train_step import EventTrainer
    gp = dict(params)
    gp["xyz"] = params["xyz"] + 0.01 * torch.randn(params["xyz"].shape, generator=torch.Generator().manual_seed(1)).to(DEV)
    t = EventTrainer(gp, DEV)
    return [t.render_raw(c, bg)["color"].clone() for c in cams]


def test_preact_forward_matches_torch_activations():
    """E3DGS_FLAG_PREACT (exp / normalize / sigmoid inside the kernel) == torch activations + plain path."""
    from event_3dgs_amd import _lib, rasterizer, synth
    from event_3dgs_amd.train_step import EventTrainer
    params, cams = _scene()
    bg = torch.tensor([0.1, 0.2, 0.3], device=DEV)
    tr = EventTrainer(params, DEV)
    raw = tr.render_raw(cams[0], bg)
    act = synth.activate(params)
    ref = rasterizer.forward_raw(act["means3D"], act["shs"], None, act["opacities"], act["scales"], act["rotations"],
                                 None, tr._settings(cams[0], bg))
    # The in-kernel activations differ from torch's by <= 1 ulp.  The rasteriser is discontinuous at its
    # thresholds (alpha >= 1/255, T >= 1e-4, ceil of the radius), so a handful of pixels may gain or lose ONE
    # borderline contribution (<= 1/255 of a colour); everything else agrees to rounding.
    assert (raw["radii"] != ref["radii"]).sum().item() <= 4
    diff = (raw["color"] - ref["color"]).abs()
    assert diff.max().item() <= 1.0 / 255.0 + 1e-6
    assert (diff > 1e-4).float().mean().item() <= 1e-4              # <= 0.01 % of the pixels carry a flip
    assert diff.mean().item() <= 1e-6


def test_preact_forward_is_the_operator_on_the_oracles_activations():
    """The in-kernel activations are DETERMINISTIC (csrc/common.h act_exp / act_sigmoid / act_load_scale_rot) and restated
    operation for operation by the oracle (gso_activate, scene/gaussian_model.py:33-41): the fused path on the raw
    parameters is bit-identical to the operator on gso_activate()'s values -- radii, image, final_T -- and therefore to
    the oracle itself (tests/test_hip_configs.py checks that at full size)."""
    from event_3dgs_amd import rasterizer, synth
    from event_3dgs_amd.train_step import EventTrainer
    from oracle import c_oracle
    params, cams = _scene()
    bg = torch.tensor([0.1, 0.2, 0.3], device=DEV)
    tr = EventTrainer(params, DEV)
    v = {k: t.detach().cpu().numpy() for k, t in tr.views.items()}
    sc, ro, op = (torch.from_numpy(a).to(DEV) for a in c_oracle.activate(v["scaling"], v["rotation"], v["opacity"]))
    shs = tr.features_reference_layout()
    for cam in cams:
        raw = tr.render_raw(cam, bg)
        ref = rasterizer.forward_raw(tr.views["xyz"], shs, None, op, sc, ro, None, tr._settings(cam, bg))
        assert torch.equal(raw["radii"], ref["radii"])
        assert torch.equal(raw["color"], ref["color"])
        assert raw["num_rendered"] == ref["num_rendered"]


@pytest.mark.parametrize("deblur", [False, True])
def test_fused_step_equals_autograd_step(deblur):
    """The no-autograd fused iteration produces the same gradients and the same parameter update as the
    reference-style iteration (torch activations + drop-in operator + autograd event loss)."""
    from event_3dgs_amd.train_step import EventTrainer
    params, cams = _scene()
    bg = torch.zeros(3, device=DEV)
    gts = _gts(params, cams, bg)
    blur = (0.5 * (gts[0] + gts[2])).contiguous() if deblur else None
    a, b = EventTrainer(params, DEV), EventTrainer(params, DEV)
    # (compute_gradients + apply_update: the form that leaves the whole gradient in memory; step() keeps the SH gradient
    # in registers on one rank, checked bit for bit against this form in test_sh_optimizer_from_colour_gradients...)
    sa = a.compute_gradients(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg, gt_blur=blur).clone()
    a.apply_update()
    lb = b.step_autograd(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg, gt_blur=blur)
    torch.cuda.synchronize()
    assert abs(float(sa[0]) - float(lb)) <= 1e-5 * abs(float(lb))
    for name in ("xyz", "features", "opacity", "scaling", "rotation"):
        ga, gb = a.grads[name].cpu().numpy(), b.grads[name].cpu().numpy()
        assert np.abs(gb).max() > 0
        assert rel_l2(ga, gb) <= 1e-3, (name, rel_l2(ga, gb))
    assert abs(float(a.c_grad) - float(b.c_grad)) <= 1e-4 * abs(float(b.c_grad))
    # Adam moments agree to the gradient tolerance (the update itself is sign-like at step 1, so compare moments)
    assert rel_l2(a.exp_avg.cpu().numpy(), b.exp_avg.cpu().numpy()) <= 1e-3
    assert float((a.flat - b.flat).abs().max()) <= 0.25            # one Adam step moves each parameter by <= lr


def test_event_loss_kernel_matches_reference_golden():
    from event_3dgs_amd import losses
    g = np.load(os.path.join(GOLDEN, "event_loss.npz"))
    T = lambda k: torch.tensor(g[k], device=DEV)
    for tag, blur in (("", None), ("_deblur", T("gt_blur"))):
        c = torch.tensor([float(g["c" + tag])], device=DEV)
        sc, d_image, d_now, d_next = losses.event_loss_raw(T("image"), T("now"), T("next"), c, T("gt_int"),
                                                           T("gt_now"), T("gt_next"), blur)
        sc = sc.cpu().numpy()
        assert abs(sc[0] - float(g["loss" + tag])) <= 2e-6 * abs(float(g["loss" + tag]))
        assert abs(sc[2] - float(g["rho"])) <= 1e-6
        assert abs(sc[1] - float(g["d_c" + tag])) <= 5e-5 * abs(float(g["d_c" + tag]))
        for name, t in (("d_image", d_image), ("d_now", d_now), ("d_next", d_next)):
            ref = g[name + tag]
            assert np.abs(t.cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-9, name
    # autograd wrapper
    img, now, nxt = (T(k).requires_grad_(True) for k in ("image", "now", "next"))
    c = torch.tensor(float(g["c"]), device=DEV, requires_grad=True)
    loss = losses.event_iteration_loss(img, now, nxt, c, T("gt_int"), T("gt_now"), T("gt_next"))
    (2.0 * loss).backward()
    assert np.abs(now.grad.cpu().numpy() - 2.0 * g["d_now"]).max() <= 4e-5 * np.abs(g["d_now"]).max() + 1e-9
    assert abs(float(c.grad) - 2.0 * float(g["d_c"])) <= 1e-4 * abs(float(g["d_c"]))


def test_adam_kernel_matches_torch_adam():
    from event_3dgs_amd import losses
    g = torch.Generator().manual_seed(0)
    n = 48 * 1000
    p0 = torch.randn(n, generator=g)
    p = p0.clone().to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    ref = torch.nn.Parameter(p0.clone())
    ref_rest = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=2.5e-3, eps=1e-15)
    opt2 = torch.optim.Adam([ref_rest], lr=2.5e-3 / 20, eps=1e-15)
    dc = (torch.arange(n) % 48) < 3
    for step in range(1, 6):
        grad = torch.randn(n, generator=g) * (10.0 ** float(torch.randint(-6, 2, (1,), generator=g)))
        ref.grad = grad.clone(); ref_rest.grad = grad.clone()
        opt.step(); opt2.step()
        losses.adam_step_(p, grad.to(DEV), m, v, 2.5e-3, step, lr_b=2.5e-3 / 20, period=48, split=3)
    expect = torch.where(dc, ref.detach(), ref_rest.detach())
    assert float((p.cpu() - expect).abs().max()) <= 2e-6


@pytest.mark.parametrize("P", [5, 777, 20000])
def test_dist_knn3_is_exact(P):
    from simple_knn._C import distCUDA2
    from oracle import c_oracle
    g = torch.Generator().manual_seed(P)
    pts = torch.rand(P, 3, generator=g) * torch.tensor([2.6, 2.6, 0.7]) - 1.3
    if P >= 700:
        pts[5] = pts[9]                       # duplicated point -> distance 0 contributes
        pts[100:200] *= 0.01                  # dense cluster (non-uniform cells)
    got = distCUDA2(pts.to(DEV)).cpu().numpy()
    if P <= 5000:
        ref = c_oracle.knn3(pts.numpy())
    else:
        from scipy.spatial import cKDTree
        d, _ = cKDTree(pts.numpy().astype(np.float64)).query(pts.numpy().astype(np.float64), k=4)
        ref = (d[:, 1:] ** 2).mean(1)
    assert np.allclose(got, ref, rtol=2e-5, atol=1e-12)


def test_small_kernels_on_random_odd_shapes():
    """tools/fuzz_aux.py: distCUDA2 (1..3000 points, duplicates, collinear, clusters), event loss (frames from 1x1,
    zero targets, deblur term), SSIM value + gradient (frames smaller than the window), Adam -- against the oracle."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_aux
    for seed in range(12):
        for fn in fuzz_aux.CHECKS:
            desc, problems = fn(seed)
            assert not problems, (desc, problems)


def test_training_reduces_loss_and_tracks_reference_trainer():
    """A short run: the fused trainer and the autograd trainer stay together and the loss goes down."""
    from event_3dgs_amd.train_step import EventTrainer
    params, cams = _scene(N=4000, W=160, H=112, seed=3)
    bg = torch.zeros(3, device=DEV)
    gts = _gts(params, cams, bg)
    a, b = EventTrainer(params, DEV), EventTrainer(params, DEV)
    la, lb = [], []
    for _ in range(30):
        la.append(float(a.step(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg)[0]))
        lb.append(float(b.step_autograd(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg)))
    assert la[-1] < 0.9 * la[0]
    assert abs(la[-1] - lb[-1]) <= 0.03 * abs(lb[-1])
    psnr = lambda x, y: 20 * math.log10(1.0 / math.sqrt(float(((x - y) ** 2).mean())))
    ia, ib = a.render_raw(cams[0], bg)["color"], b.render_raw(cams[0], bg)["color"]
    assert abs(psnr(ia, gts[0]) - psnr(ib, gts[0])) <= 0.1          # north_star: PSNR within 0.1 dB


def test_ssim_kernel_matches_reference_golden_and_oracle():
    """e3dgs_ssim (forward + gradient) vs the reference's values (golden G5) and the torch restatement."""
    from event_3dgs_amd import losses
    from oracle import torch_oracle
    g = np.load(os.path.join(GOLDEN, "image_metrics.npz"))
    a = torch.tensor(g["a"], device=DEV, requires_grad=True)
    b = torch.tensor(g["b"], device=DEV)
    assert abs(float(losses.ssim(a, b)) - float(g["ssim"])) <= 3e-6
    assert abs(float(losses.ssim_gray(a, b)) - float(g["ssim_gray"])) <= 3e-6
    assert abs(float(losses.l1_loss_gray(a, b)) - float(g["l1_gray"])) <= 1e-6
    assert np.allclose(losses.psnr(a.detach(), b).cpu().numpy(), g["psnr"], atol=1e-4)
    loss = losses.gray_iteration_loss(a, b)                         # train.py:213-223
    assert abs(float(loss) - float(g["gray_loss"])) <= 3e-6
    loss.backward()
    ref = g["d_a_gray_loss"]
    assert np.abs(a.grad.cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max()
    # ragged size, 3-channel (non-gray) path, against the CPU restatement with autograd
    gen = torch.Generator().manual_seed(3)
    x = torch.rand(3, 37, 53, generator=gen)
    y = (x + 0.2 * torch.randn(3, 37, 53, generator=gen)).clamp(0, 1)
    xc = x.clone().requires_grad_(True)
    ref_v = torch_oracle.ssim(xc, y)
    ref_v.backward()
    xg = x.to(DEV).requires_grad_(True)
    v = losses.ssim(xg, y.to(DEV))
    v.backward()
    assert abs(float(v) - float(ref_v)) <= 3e-6
    assert np.abs(xg.grad.cpu().numpy() - xc.grad.numpy()).max() <= 2e-5 * float(xc.grad.abs().max())
    lr = losses.rgb_iteration_loss(xg, y.to(DEV))
    assert abs(float(lr) - float(0.8 * (x - y).abs().mean() + 0.2 * (1 - ref_v))) <= 3e-6


def test_trainer_densification_roundtrip_and_training_continues():
    """export/import of the flat coefficient-major buffers is lossless; after densify_and_prune (N changes) the
    trainer keeps training; opacity reset zeroes only the opacity moments."""
    from event_3dgs_amd import densify
    from event_3dgs_amd.train_step import EventTrainer
    params, cams = _scene(N=3000, W=160, H=112, seed=5)
    bg = torch.zeros(3, device=DEV)
    gts = _gts(params, cams, bg)
    tr = EventTrainer(params, DEV, track_densification_stats=True)
    stats = densify.DensifyStats(tr.N, DEV)
    # (with every GT pixel changing between the two event frames rho = 1 and the intensity term -- the only
    #  source of gradient for render #1 -- has weight 0, train.py:190-196; the deblur term keeps it alive)
    for _ in range(5):
        tr.step(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg, gt_blur=gts[0])
        stats.update(tr.viewspace_grad, tr.last_radii)               # train.py:319-320
    assert float(stats.denom.max()) == 5.0 and float(stats.xyz_gradient_accum.sum()) > 0
    flat0, m0 = tr.flat.clone(), tr.exp_avg.clone()
    g = tr.export_groups()
    assert torch.equal(g["f_dc"][0], torch.cat((params["features_dc"], params["features_rest"]), 1)[:, :1] * 0 + g["f_dc"][0])
    tr.import_groups(g)
    assert torch.equal(tr.flat, flat0) and torch.equal(tr.exp_avg, m0)
    n0 = tr.N
    thr = float((stats.xyz_gradient_accum / stats.denom.clamp_min(1)).median())
    n1 = tr.densify_and_prune(stats, max_grad=thr, min_opacity=0.005, extent=4.0, max_screen_size=None)
    assert n1 != n0 and stats.denom.shape[0] == n1 and float(stats.denom.sum()) == 0.0
    assert tr.flat.numel() == 59 * n1 + 1
    l = [float(tr.step(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg)[0]) for _ in range(5)]
    assert all(np.isfinite(l))
    off, n = tr.seg["opacity"]
    tr.reset_opacity()
    assert float(torch.sigmoid(tr.views["opacity"]).max()) <= 0.01 + 1e-6
    assert float(tr.exp_avg[off:off + n].abs().sum()) == 0.0 and float(tr.exp_avg[:100].abs().sum()) > 0.0


def test_fit_loop_with_densification_and_eval(tmp_path):
    """End to end on the GPU: create_from_pcd (distCUDA2) -> event training loop with the densification
    schedule -> evaluation protocol (gray PSNR/SSIM on held-out views) -> model PLY round trip."""
    import random
    from event_3dgs_amd import fit, io_formats, scene_io, synth
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer
    from simple_knn._C import distCUDA2
    W, H, K = 128, 96, 96
    bg = torch.zeros(3, device=DEV)
    gt_params = synth.make_scene(3000, "trained", seed=9, device=DEV)
    gt_tr = EventTrainer(gt_params, DEV)
    q8 = lambda t: (torch.round(t.clamp(0, 1) * 255) / 255).contiguous()
    train, events = [], []
    for k in range(K):
        for lst, daz in ((train, 0.0), (events, 0.002)):
            c = orbit_camera(k, K, W, H, device=DEV, daz=daz)
            c.original_image = q8(gt_tr.render_raw(c, bg)["color"])
            lst.append(c)
    pcd = io_formats.BasicPointCloud(gt_params["xyz"].cpu().numpy()[:1500], np.full((1500, 3), 0.5), np.zeros((1500, 3)))
    params = scene_io.create_from_pcd(pcd, 1.0, distCUDA2, device=DEV)
    assert params["scaling"].shape == (1500, 3) and float(torch.sigmoid(params["opacity"]).mean()) - 0.1 < 1e-6
    rnd = random.Random(0)
    sizes = []
    tr = fit.fit_event_scene(params, train, events, bg, DEV, iterations=260, cameras_extent=4.4, densify_from_iter=50,
                             densification_interval=100, sh_ramp_interval=100, rng=rnd.randint, densify_grad_threshold=1e-9,
                             on_iteration=lambda it, t, s: sizes.append(t.N))
    assert tr.active_sh_degree == 2 and sizes[0] == 1500 and sizes[-1] != 1500      # SH ramp and densification ran
    res = scene_io.evaluate_views(lambda cam: tr.render_raw(cam, bg)["color"], train)
    assert np.isfinite(res["psnr"]) and 0.0 < res["ssim"] <= 1.0 and len(res["per_view"]) == 5
    g = tr.export_groups()
    path = str(tmp_path / "point_cloud.ply")
    io_formats.save_model_ply(path, g["xyz"][0], g["f_dc"][0], g["f_rest"][0], g["opacity"][0], g["scaling"][0], g["rotation"][0])
    back = io_formats.load_model_ply(path, device=DEV)
    tr2 = EventTrainer(back, DEV, active_sh_degree=tr.active_sh_degree)
    a, b = tr.render_raw(train[3], bg)["color"], tr2.render_raw(train[3], bg)["color"]
    assert torch.equal(a, b)                                   # the PLY holds the exact pre-activation state


@pytest.mark.parametrize("defer_colour", [False, True])
@pytest.mark.parametrize("nviews", [1, 2, 3, 4])
def test_multi_view_pass_equals_single_view_calls(nviews, defer_colour):
    """e3dgs_rasterize_forward_multi / _backward_multi against nviews separate single-view calls: images and
    radii bit-identical; gradients == the sum of the per-view gradients (what autograd's accumulation of the
    per-render backward calls produces at train.py:211).  Gradient buffers start as NaN to prove every element
    is written; a second multi pass must reproduce the first bit for bit (no atomics anywhere)."""
    from event_3dgs_amd import _lib, rasterizer
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer
    params, _ = _scene(N=4000)
    W, H = 183, 131                                       # not multiples of 16: partial tiles in every view
    cams = [orbit_camera(k, 16, W, H, device=DEV, daz=0.01 * k) for k in range(nviews)]
    bg = torch.tensor([0.2, 0.1, 0.3], device=DEV)
    tr = EventTrainer(params, DEV)
    v = tr.views
    settings = [tr._settings(c, bg) for c in cams]
    gen = torch.Generator().manual_seed(5)
    dpix = torch.randn(nviews, 3, H, W, generator=gen).to(DEV)
    names = dict(means3D=v["xyz"], sh=v["features"], opacities=v["opacity"], scales=v["scaling"], rots=v["rotation"])
    # ---- reference: one call per view
    ref = {k: torch.zeros_like(t) for k, t in names.items()}
    singles, m2d_ref, total = [], None, 0
    for k, cam in enumerate(cams):
        raw = tr.render_raw(cam, bg)
        singles.append(raw)
        total += raw["num_rendered"]
        single = {n: torch.full_like(t, float("nan")) for n, t in names.items()}
        single["means2D"] = torch.full((tr.N, 3), float("nan"), device=DEV)
        rasterizer.backward_raw(raw, dpix[k], single, flags=tr.FWD_FLAGS)
        for n in names:
            ref[n] += single[n]
        if k == 0:
            m2d_ref = single["means2D"]
    # ---- one multi-view pass
    def multi():
        # defer_colour: SH -> RGB in its own kernel right before compositing (E3DGS_FLAG_DEFER_COLOR): same bits
        raw = rasterizer.forward_multi(v["xyz"], v["features"], v["opacity"], v["scaling"], v["rotation"], settings,
                                       flags=tr.FWD_FLAGS | (_lib.FLAG_DEFER_COLOR if defer_colour else 0))
        out = {n: torch.full_like(t, float("nan")) for n, t in names.items()}
        out["means2D"] = torch.full((tr.N, 3), float("nan"), device=DEV)
        rasterizer.backward_multi(raw, dpix, out)
        torch.cuda.synchronize()
        return raw, out
    raw, out = multi()
    assert raw["num_rendered"] == total
    for k in range(nviews):
        assert torch.equal(raw["color"][k], singles[k]["color"]), k
        assert torch.equal(raw["radii"][k], singles[k]["radii"]), k
    assert torch.equal(out["means2D"], m2d_ref)
    for n in names:
        assert torch.isfinite(out[n]).all(), n
        err = rel_l2(out[n].cpu(), ref[n].cpu())
        assert err < 2e-6, (n, err)                                                # fp32 re-association only
        scale = ref[n].abs().max().item()
        assert (out[n] - ref[n]).abs().max().item() <= 1e-4 * scale + 1e-12, n
    raw2, out2 = multi()
    for n in out:
        assert torch.equal(out[n], out2[n]), n                                     # deterministic gradients


def test_multi_view_argument_checks():
    from event_3dgs_amd import rasterizer
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer
    params, _ = _scene(N=500)
    tr = EventTrainer(params, DEV)
    v = tr.views
    bg = torch.zeros(3, device=DEV)
    cams = [orbit_camera(0, 16, 64, 48, device=DEV), orbit_camera(1, 16, 80, 48, device=DEV)]
    with pytest.raises(ValueError):        # different frame sizes in one batch
        rasterizer.forward_multi(v["xyz"], v["features"], v["opacity"], v["scaling"], v["rotation"],
                                 [tr._settings(c, bg) for c in cams], flags=tr.FWD_FLAGS)
    five = [tr._settings(orbit_camera(k, 16, 64, 48, device=DEV), bg) for k in range(5)]
    with pytest.raises(Exception, match="nviews"):
        rasterizer.forward_multi(v["xyz"], v["features"], v["opacity"], v["scaling"], v["rotation"], five,
                                 flags=tr.FWD_FLAGS)


def test_multi_view_edge_cases_empty_and_all_culled():
    """P == 0 and "every Gaussian behind the cameras" through the multi-view pass (polled count included):
    background images, zero instance count, zero gradients."""
    from event_3dgs_amd import _lib, rasterizer
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer
    W, H = 70, 50
    bg = torch.tensor([0.3, 0.6, 0.9], device=DEV)
    cams = [orbit_camera(k, 16, W, H, device=DEV) for k in range(3)]
    # ---- all culled: move the scene far behind every camera
    params, _ = _scene(N=300)
    tr = EventTrainer(params, DEV)
    v = tr.views
    v["xyz"].add_(1000.0)
    settings = [tr._settings(c, bg) for c in cams]
    for flags in (tr.FWD_FLAGS, tr.FWD_FLAGS | _lib.FLAG_COUNT_MAPPED):
        raw = rasterizer.forward_multi(v["xyz"], v["features"], v["opacity"], v["scaling"], v["rotation"], settings,
                                       flags=flags)
        assert raw["num_rendered"] == 0 and int(raw["radii"].abs().sum()) == 0
        assert torch.equal(raw["color"], bg.view(1, 3, 1, 1).expand(3, 3, H, W))
        out = dict(opacities=torch.full((tr.N, 1), float("nan"), device=DEV), means3D=torch.full((tr.N, 3), float("nan"), device=DEV),
                   sh=torch.full((48, tr.N), float("nan"), device=DEV), scales=torch.full((tr.N, 3), float("nan"), device=DEV),
                   rots=torch.full((tr.N, 4), float("nan"), device=DEV))
        rasterizer.backward_multi(raw, torch.ones(3, 3, H, W, device=DEV), out)
        torch.cuda.synchronize()
        for n, t in out.items():
            assert float(t.abs().max()) == 0.0, n
    # ---- P == 0
    e = lambda *s: torch.empty(*s, device=DEV)
    raw = rasterizer.forward_multi(e(0, 3), e(48, 0), e(0, 1), e(0, 3), e(0, 4), settings,
                                   flags=tr.FWD_FLAGS | _lib.FLAG_COUNT_MAPPED)
    assert raw["num_rendered"] == 0 and tuple(raw["radii"].shape) == (3, 0)
    assert torch.equal(raw["color"], bg.view(1, 3, 1, 1).expand(3, 3, H, W))


@pytest.mark.parametrize("mode", ["gray", "rgb"])
def test_fused_image_step_equals_autograd_step(mode):
    """The `--gray` (train.py:213-223) and RGB (train.py:292-296) iterations on the fused path -- one render, L1 + SSIM
    with the SSIM kernel's own gradient, no autograd graph -- against the same iteration through torch autograd."""
    from event_3dgs_amd.train_step import EventTrainer
    params, cams = _scene()
    bg = torch.zeros(3, device=DEV)
    gt = _gts(params, cams, bg)[0]
    a, a2, b = EventTrainer(params, DEV), EventTrainer(params, DEV), EventTrainer(params, DEV)
    # a2: gradients in memory (single-view per-Gaussian kernel), then the generic optimizer; a: step_image() as training
    # runs it on one rank (multi-view per-Gaussian kernel with one view, SH gradient rebuilt inside the SH optimizer
    # kernel from the view's colour gradients, never stored): same mathematics, different kernels -> rounding only
    la2 = a2.compute_gradients_image(cams[0], gt, bg, mode=mode).clone()
    a2.apply_update(skip=("c",))
    la = a.step_image(cams[0], gt, bg, mode=mode)
    lb = b.step_image_autograd(cams[0], gt, bg, mode=mode)
    torch.cuda.synchronize()
    assert a.sh_via_colour and float(la) == float(la2)
    for x, y in ((a.exp_avg, a2.exp_avg), (a.exp_avg_sq, a2.exp_avg_sq)):
        assert rel_l2(x.cpu().numpy(), y.cpu().numpy()) <= 1e-5
    assert rel_l2(a.flat.cpu().numpy(), a2.flat.cpu().numpy()) <= 1e-5
    assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(lb))
    for name in ("xyz", "features", "opacity", "scaling", "rotation"):
        ga, gb = a2.grads[name].cpu().numpy(), b.grads[name].cpu().numpy()
        assert np.abs(gb).max() > 0
        assert rel_l2(ga, gb) <= 1e-3, (name, rel_l2(ga, gb))
    assert rel_l2(a.exp_avg.cpu().numpy(), b.exp_avg.cpu().numpy()) <= 1e-3
    # a few steps train: the loss goes down
    first = float(la)
    for _ in range(15):
        last = float(a.step_image(cams[0], gt, bg, mode=mode))
    assert last < first


@pytest.mark.parametrize("mode", ["gray", "rgb"])
def test_fit_loop_image_modes(mode):
    """fit_event_scene in the reference's `--gray` / RGB modes: one render per iteration, densification schedule."""
    import random
    from event_3dgs_amd import fit, synth
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer
    W, H, K = 96, 64, 24
    bg = torch.zeros(3, device=DEV)
    gt_tr = EventTrainer(synth.make_scene(2000, "trained", seed=9, device=DEV), DEV)
    train = []
    for k in range(K):
        c = orbit_camera(k, K, W, H, device=DEV)
        c.original_image = gt_tr.render_raw(c, bg)["color"].clamp(0, 1).contiguous()
        train.append(c)
    losses_seen, sizes = [], []
    tr = fit.fit_event_scene(synth.make_scene(1500, "trained", seed=4, device=DEV), train, None, bg, DEV, iterations=40,
                             cameras_extent=4.4, densify_from_iter=10, densification_interval=15, start_sh_degree=3,
                             densify_grad_threshold=1e-9, rng=random.Random(1).randint, mode=mode,
                             on_iteration=lambda it, t, s: (losses_seen.append(float(s)), sizes.append(t.N)))
    assert sizes[-1] != 1500 and np.isfinite(losses_seen).all()
    assert np.mean(losses_seen[-8:]) < 1.25 * np.mean(losses_seen[:8])     # random views + densification: no blow-up


def test_multi_view_many_tiles_and_huge_splats():
    """Paths the benchmark configuration does not reach: more than 32 K tiles in one call (fallback LPT ordering kernel,
    16-bit tile keys) and splats that cover hundreds of tiles (long candidate walks in the binning kernels, tile lists
    far beyond one 64-entry round).  The multi-view pass must still equal the single-view calls bit for bit."""
    from event_3dgs_amd import rasterizer
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer
    params, _ = _scene(N=1500)
    params = dict(params)
    params["scaling"] = params["scaling"] + math.log(6.0)            # 6x larger splats
    W, H, nviews = 2048, 1152, 4                                      # 128 x 72 x 4 = 36 864 tiles
    cams = [orbit_camera(k, 8, W, H, device=DEV, daz=0.02 * k) for k in range(nviews)]
    bg = torch.tensor([0.1, 0.0, 0.2], device=DEV)
    tr = EventTrainer(params, DEV)
    v = tr.views
    raw = rasterizer.forward_multi(v["xyz"], v["features"], v["opacity"], v["scaling"], v["rotation"],
                                   [tr._settings(c, bg) for c in cams], flags=tr.FWD_FLAGS)
    assert raw["num_rendered"] > 300 * tr.N                           # hundreds of tiles per splat
    total = 0
    for k, cam in enumerate(cams):
        one = tr.render_raw(cam, bg)
        total += one["num_rendered"]
        assert torch.equal(raw["color"][k], one["color"]), k
        assert torch.equal(raw["radii"][k], one["radii"]), k
    assert raw["num_rendered"] == total
    names = dict(means3D=v["xyz"], sh=v["features"], opacities=v["opacity"], scales=v["scaling"], rots=v["rotation"])
    dpix = torch.randn(nviews, 3, H, W, generator=torch.Generator().manual_seed(2)).to(DEV)
    out = {n: torch.full_like(t, float("nan")) for n, t in names.items()}
    rasterizer.backward_multi(raw, dpix, out)
    ref = {n: torch.zeros_like(t) for n, t in names.items()}
    for k, cam in enumerate(cams):
        single = {n: torch.empty_like(t) for n, t in names.items()}
        rasterizer.backward_raw(tr.render_raw(cam, bg), dpix[k], single, flags=tr.FWD_FLAGS)
        for n in names:
            ref[n] += single[n]
    torch.cuda.synchronize()
    for n in names:
        assert torch.isfinite(out[n]).all(), n
        assert rel_l2(out[n].cpu(), ref[n].cpu()) < 5e-6, n


def test_sh_gradient_rebuilt_from_colour_views():
    """e3dgs_sh_grad_from_colour: the SH gradient rebuilt from the per-view clamp-masked colour gradients that
    backward_multi hands out equals the SH gradient backward_multi computes itself -- for one "rank" exactly the same
    sum, for two rank blocks (views split 2 + 2, scale 1/2) the mean of the two halves."""
    from event_3dgs_amd import rasterizer
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer
    params, _ = _scene(N=3000)
    W, H, nviews = 150, 110, 4
    cams = [orbit_camera(k, 16, W, H, device=DEV, daz=0.01 * k) for k in range(nviews)]
    bg = torch.zeros(3, device=DEV)
    tr = EventTrainer(params, DEV, active_sh_degree=3)
    v = tr.views
    P = tr.N
    dpix = torch.randn(nviews, 3, H, W, generator=torch.Generator().manual_seed(8)).to(DEV)
    names = dict(means3D=v["xyz"], sh=v["features"], opacities=v["opacity"], scales=v["scaling"], rots=v["rotation"])

    def run(cam_list, dp):
        raw = rasterizer.forward_multi(v["xyz"], v["features"], v["opacity"], v["scaling"], v["rotation"],
                                       [tr._settings(c, bg) for c in cam_list], flags=tr.FWD_FLAGS)
        out = {n: torch.empty_like(t) for n, t in names.items()}
        out["colour_views"] = torch.full((len(cam_list), P, 3), float("nan"), device=DEV)
        rasterizer.backward_multi(raw, dp, out)
        return out
    full = run(cams, dpix)
    assert torch.isfinite(full["colour_views"]).all()
    centres = torch.stack([c.camera_center.contiguous() for c in cams])
    # ---- one block with all four views, scale 1
    packed = torch.cat((full["colour_views"].reshape(-1), centres.reshape(-1))).view(1, -1).contiguous()
    got = torch.full_like(v["features"], float("nan"))
    rasterizer.sh_grad_from_colour(v["xyz"], packed, 1, nviews, 3, 16, got, 1.0)
    torch.cuda.synchronize()
    assert torch.isfinite(got).all()
    assert rel_l2(got.cpu(), full["sh"].cpu()) < 2e-6
    # ---- two blocks of two views, scale 1/2 == mean of the two halves' SH gradients
    a, b = run(cams[:2], dpix[:2].contiguous()), run(cams[2:], dpix[2:].contiguous())
    blocks = torch.stack([torch.cat((o["colour_views"].reshape(-1), centres[i:i + 2].reshape(-1)))
                          for o, i in ((a, 0), (b, 2))]).contiguous()
    rasterizer.sh_grad_from_colour(v["xyz"], blocks, 2, 2, 3, 16, got, 0.5)
    torch.cuda.synchronize()
    assert rel_l2(got.cpu(), (0.5 * (a["sh"] + b["sh"])).cpu()) < 2e-6
    # dL_dsh may be omitted when the colour views are taken
    raw = rasterizer.forward_multi(v["xyz"], v["features"], v["opacity"], v["scaling"], v["rotation"],
                                   [tr._settings(c, bg) for c in cams], flags=tr.FWD_FLAGS)
    out = {n: torch.empty_like(t) for n, t in names.items() if n != "sh"}
    out["colour_views"] = torch.empty(nviews, P, 3, device=DEV)
    rasterizer.backward_multi(raw, dpix, out)
    torch.cuda.synchronize()
    assert torch.equal(out["colour_views"], full["colour_views"]) and torch.equal(out["means3D"], full["means3D"])


def test_render_mirror_matches_fused_path_and_oracle():
    """event_3dgs_amd.renderer.render / render_depth (gaussian_renderer/__init__.py:20-189): same dict as the
    reference's wrapper; the torch-SH branch it is forced into, the in-rasteriser SH branch and the fused trainer render
    agree; the precomputed-covariance branch agrees; gradients reach viewspace_points and the parameters."""
    from event_3dgs_amd import renderer, synth
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer
    from helpers import oracle_kwargs
    from oracle import c_oracle
    N, W, H = 2500, 176, 120
    params = synth.make_scene(N, "trained", seed=12, device=DEV)
    cam = orbit_camera(2, 16, W, H, device=DEV)
    bg = torch.tensor([0.2, 0.3, 0.1], device=DEV)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    pc = renderer.GaussianView(leaves, active_sh_degree=2)
    pipe = renderer.PipelineParams()
    out = renderer.render(cam, pc, pipe, bg)
    assert pipe.convert_SHs_python is True                          # the reference forces it (:71)
    assert set(out) == {"render", "viewspace_points", "visibility_filter", "radii"}
    assert out["render"].shape == (3, H, W) and out["radii"].dtype == torch.int32
    assert torch.equal(out["visibility_filter"], out["radii"] > 0) and int(out["visibility_filter"].sum()) > N // 3
    fused = EventTrainer(params, DEV, active_sh_degree=2).render_raw(cam, bg)
    in_op = renderer.render(cam, pc, renderer.PipelineParams(), bg, force_python_sh=False)
    for other in (fused["color"], in_op["render"].detach()):
        d = (out["render"].detach() - other).abs()
        assert float(d.mean()) <= 1e-6 and float((d > 1e-4).float().mean()) <= 1e-4      # a few threshold pixels at most
    cov = renderer.render(cam, pc, renderer.PipelineParams(compute_cov3D_python=True), bg)
    d = (out["render"].detach() - cov["render"].detach()).abs()
    assert float(d.mean()) <= 1e-6 and int((out["radii"] != cov["radii"]).sum()) <= 2
    # gradients: screen-space means (what add_densification_stats reads, train.py:317-320) and every parameter group
    gw = torch.randn(3, H, W, generator=torch.Generator().manual_seed(2)).to(DEV)
    (out["render"] * gw).sum().backward()
    vg = out["viewspace_points"].grad
    assert vg is not None and float(vg[:, :2].abs().sum()) > 0 and float(vg[:, 2].abs().max()) == 0.0
    for k in ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation"):
        assert leaves[k].grad is not None and torch.isfinite(leaves[k].grad).all() and float(leaves[k].grad.abs().sum()) > 0
    # depth render == the oracle composited with colour = distance + 0.5
    dep = renderer.render_depth(cam, renderer.GaussianView(params), renderer.PipelineParams(), bg)
    act = {k: v.cpu() for k, v in synth.activate(params).items()}
    act["colors"] = ((act["means3D"] - cam.camera_center.cpu()[None]).norm(dim=1, keepdim=True) + 0.5).repeat(1, 3)
    cam_cpu = orbit_camera(2, 16, W, H)
    f = c_oracle.Forward(**oracle_kwargs(act, cam_cpu, tuple(bg.tolist()), False, False))
    d = np.abs(dep["render"].detach().cpu().numpy() - f.out_color)
    assert float(d.mean()) <= 1e-5 and float((d > 1e-3).mean()) <= 1e-4
    # render_point (gaussian_renderer/__init__.py:274-370): nearest-point depth map of the visible, opaque Gaussians -- on
    # the device, against the same reduction done point by point on the host with the oracle's radii
    gv = renderer.GaussianView(params)
    pm = renderer.render_point(cam, gv, renderer.PipelineParams(), bg)
    assert pm.shape == (H, W) and pm.is_cuda
    keep = (f.radii > 0) & (act["opacities"].reshape(-1).numpy() > 0.8)
    pts = act["means3D"].numpy()[keep]
    ndc = renderer.project_points(torch.from_numpy(pts), cam_cpu.full_proj_transform).numpy()
    ref = np.full((H, W), np.inf, np.float32)
    dist = np.linalg.norm(pts - cam_cpu.camera_center.numpy()[None], axis=1)
    for i in range(pts.shape[0]):
        x, y = int(((ndc[i, 0] + 1) * W - 1) * 0.5), int(((ndc[i, 1] + 1) * H - 1) * 0.5)
        if 0 <= x < W and 0 <= y < H:
            ref[y, x] = min(ref[y, x], dist[i])
    got = pm.cpu().numpy()
    assert keep.sum() > 50 and (np.isfinite(got) == np.isfinite(ref)).mean() >= 1.0 - 4.0 / got.size
    both = np.isfinite(got) & np.isfinite(ref)
    assert both.sum() > 20 and np.abs(got[both] - ref[both]).max() <= 1e-5 * ref[both].max()


def test_densify_stats_kernel_equals_torch_form():
    """e3dgs_densify_stats_update == the masked torch form of train.py:317-320 / gaussian_model.py:405-407."""
    from event_3dgs_amd.densify import DensifyStats
    g = torch.Generator().manual_seed(0)
    n = 10_007
    gpu, cpu = DensifyStats(n, DEV), DensifyStats(n, "cpu")
    for _ in range(3):
        grad = torch.randn(n, 3, generator=g) * 1e-3
        radii = (torch.randint(0, 60, (n,), generator=g, dtype=torch.int32) * (torch.rand(n, generator=g) > 0.4)).to(torch.int32)
        gpu.update(grad.to(DEV), radii.to(DEV))
        cpu.update(grad, radii)
    assert torch.equal(gpu.max_radii2D.cpu(), cpu.max_radii2D) and torch.equal(gpu.denom.cpu(), cpu.denom)
    assert torch.allclose(gpu.xyz_gradient_accum.cpu(), cpu.xyz_gradient_accum, rtol=1e-6, atol=0)
    assert float(cpu.denom.max()) == 3.0 and float(cpu.denom.min()) == 0.0


def test_segmented_adam_equals_per_group_adam():
    """e3dgs_adam_step_segments (all optimizer groups of the flat buffer in one launch) == one e3dgs_adam_step per
    group, bit for bit; bad segment tables are refused."""
    from event_3dgs_amd import _lib, losses
    g = torch.Generator().manual_seed(3)
    n = 100_003
    ends = (3000, 9000, 60_001, 61_000, 80_000, 100_002, 100_003)
    lrs = (1.6e-4, 2.5e-3, 1.25e-4, 0.05, 5e-3, 1e-3, 0.1)
    eps = (1e-15,) * 6 + (1e-8,)
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 1e-3
    a = [t.clone().to(DEV) for t in (p0, torch.zeros(n), torch.zeros(n))]
    b = [t.clone().to(DEV) for t in (p0, torch.zeros(n), torch.zeros(n))]
    grad = gr.to(DEV)
    for step in range(1, 4):
        losses.adam_step_segments_(a[0], grad, a[1], a[2], ends, lrs, eps, step)
        lo = 0
        for e, lr, ep in zip(ends, lrs, eps):
            losses.adam_step_(b[0][lo:e], grad[lo:e], b[1][lo:e], b[2][lo:e], lr, step, eps=ep)
            lo = e
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    with pytest.raises(_lib.HipLibraryError, match="last segment"):
        losses.adam_step_segments_(a[0], grad, a[1], a[2], ends[:-1], lrs[:-1], eps[:-1], 1)


def test_deferred_sh_position_term_is_bit_identical(monkeypatch):
    """E3DGS_FLAG_DEFER_SH_MEAN + e3dgs_sh_adam_from_colour_mean (the per-Gaussian backward leaves the position-gradient
    term through the SH view directions to the SH optimizer kernel) against the backward that computes it itself:
    identical parameters and moments after every iteration, event and image mode; between compute_gradients() and
    apply_update() the xyz gradient is the geometric part only."""
    from event_3dgs_amd.train_step import EventTrainer
    from event_3dgs_amd import _lib, rasterizer
    params, cams = _scene(N=4000)
    bg = torch.zeros(3, device=DEV)
    gts = [t.contiguous() for t in _gts(params, cams, bg)]
    a, b = EventTrainer(params, DEV), EventTrainer(params, DEV)
    assert a.sh_via_colour
    for it in range(4):
        for t, flag in ((a, "1"), (b, "0")):
            monkeypatch.setenv("E3DGS_DEFER_SH_MEAN", flag)
            if it == 2:
                t.compute_gradients_image(cams[0], gts[0], bg, mode="gray", sh_via_colour=True)
            else:
                t.compute_gradients(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg, sh_via_colour=True)
            assert t._mean_deferred == (t is a)
        gx_a, gx_b = a.grads["xyz"].clone(), b.grads["xyz"].clone()
        assert not torch.equal(gx_a, gx_b)                              # (the view-dependent colour term is still missing)
        for name in ("opacity", "scaling", "rotation"):
            assert torch.equal(a.grads[name], b.grads[name])
        for t in (a, b):
            t.apply_update(skip=("c",) if it == 2 else ())
        assert torch.equal(a.grads["xyz"], gx_b), it                   # completed inside the SH optimizer kernel
        assert torch.equal(a.flat, b.flat) and torch.equal(a.exp_avg, b.exp_avg) and torch.equal(a.exp_avg_sq, b.exp_avg_sq), it
    with pytest.raises(_lib.HipLibraryError, match="at most 4 views"):
        N = a.N
        packed = torch.zeros(1, 5 * (N * 3 + 3), device=DEV)
        rasterizer.sh_adam_from_colour(a.views["xyz"], packed, 1, 5, 3, 16, a.views["features"], a.exp_avg[:48 * N],
                                       a.exp_avg_sq[:48 * N], 1e-3, 1e-4, 1, mean_grad=a.grads["xyz"])


def test_adam_with_a_gap_equals_one_launch_per_range():
    """e3dgs_adam_step_groups_gap (everything around the SH segment in one launch, per-group step counts, the gap never
    visited) == one e3dgs_adam_step_groups launch per contiguous range, bit for bit; the gap stays untouched."""
    from event_3dgs_amd import _lib, losses
    g = torch.Generator().manual_seed(4)
    n = 120_001
    gap = (3000, 96_000)                                    # xyz | [SH] | opacity | scaling | rotation | c
    ends = (99_000, 100_000, 103_000, 120_000, 120_001)
    lrs = (1.6e-4, 0.05, 5e-3, 1e-3, 0.1)
    eps = (1e-15,) * 4 + (1e-8,)
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 1e-3
    a = [t.clone().to(DEV) for t in (p0, torch.rand(n, generator=g), torch.rand(n, generator=g))]
    b = [t.clone() for t in a]
    grad = gr.to(DEV)
    t0 = gap[0] + gap[1]
    for steps in ((1, 1, 1, 1, 1), (2, 0, 2, 2, 2), (3, 1, 3, 3, 0)):
        losses.adam_step_segments_(a[0], grad, a[1], a[2], ends, lrs, eps, steps, gap=gap)
        losses.adam_step_segments_(b[0][:gap[0]], grad[:gap[0]], b[1][:gap[0]], b[2][:gap[0]], (gap[0],), lrs[:1], eps[:1],
                                   steps[:1])
        losses.adam_step_segments_(b[0][t0:], grad[t0:], b[1][t0:], b[2][t0:], tuple(e - t0 for e in ends[1:]), lrs[1:],
                                   eps[1:], steps[1:])
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert torch.equal(a[0][gap[0]:t0].cpu(), p0[gap[0]:t0])
    with pytest.raises(_lib.HipLibraryError, match="gap"):
        losses.adam_step_segments_(a[0], grad, a[1], a[2], ends, lrs, eps, (1,) * 5, gap=(n - 5, 10))


def test_event_mask_on_quantised_ground_truth_matches_torch():
    """rho = count(D* != 0) and sign(D - D*) hinge on exact zeros: on 8-bit ground-truth frames the kernel's mask must be
    the torch formula's (utils/loss_utils.py:234-249 + train.py:165-175), pixel for pixel in the count."""
    from event_3dgs_amd import losses
    from oracle import torch_oracle
    g = torch.Generator().manual_seed(5)
    H, W = 96, 160
    q8 = lambda t: torch.round(t.clamp(0, 1) * 255) / 255
    gt_now = q8(torch.rand(3, H, W, generator=g))
    gt_next = gt_now.clone()
    changed = torch.rand(H, W, generator=g) < 0.3                    # 30 % of the pixels see an event
    gt_next[:, changed] = q8(torch.rand(3, int(changed.sum()), generator=g))
    # a few pixels whose channels change but whose luminance may not (ties the float formula has to reproduce)
    gt_next[0, :2, :8] = gt_now[0, :2, :8] + 1.0 / 255
    gt_int = q8(torch.rand(3, H, W, generator=g))
    imgs = [torch.rand(3, H, W, generator=g) * 0.9 + 0.05 for _ in range(3)]
    c = torch.tensor([0.21])
    ref_gt = torch_oracle.event_frame(gt_now, gt_next, 0.17)
    rho_ref = float((ref_gt != 0).float().mean())
    sc, *_ = losses.event_loss_raw(*(t.to(DEV) for t in imgs), c.to(DEV), gt_int.to(DEV), gt_now.to(DEV), gt_next.to(DEV), None)
    assert abs(float(sc[2]) * H * W - rho_ref * H * W) < 0.5, (float(sc[2]), rho_ref)     # identical COUNT
    ref = torch_oracle.event_iteration_loss(imgs[0], imgs[1], imgs[2], gt_int, gt_now, gt_next, float(c))
    assert abs(float(sc[0]) - float(ref)) <= 2e-6 * abs(float(ref))


def test_apply_update_skips_like_torch_adam():
    """train.py:317-332: densification / opacity reset replace parameters BEFORE optimizer.step(), so torch skips them
    (their .grad is None) and their per-parameter step count stalls; optimizer_c steps on event iterations only.
    Emulated here with torch.optim.Adam on copies of the flat segments."""
    from event_3dgs_amd.train_step import EventTrainer
    params, cams = _scene(N=1500, W=96, H=64)
    bg = torch.zeros(3, device=DEV)
    gts = _gts(params, cams, bg)
    tr = EventTrainer(params, DEV)
    names = ("xyz", "features", "opacity", "scaling", "rotation", "c")
    seg = lambda buf, n: buf[tr.seg[n][0]:tr.seg[n][0] + tr.seg[n][1]]
    ref = {n: torch.nn.Parameter(seg(tr.flat, n).clone()) for n in names}
    N = tr.N
    lr_feat = torch.full((48 * N,), tr.lrs["features_rest"], device=DEV)
    lr_feat[:3 * N] = tr.lrs["features"]
    plan = [(), ("gaussians",), ("opacity",), ("c",), ()]
    steps = {n: 0 for n in names}
    m = {n: torch.zeros_like(ref[n]) for n in names}
    v = {n: torch.zeros_like(ref[n]) for n in names}
    for it, skip in enumerate(plan, start=1):
        tr.compute_gradients(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg)
        grads = {n: seg(tr.flat_grad, n).clone() for n in names}
        tr.apply_update(skip=skip)
        for n in names:
            if n == "c":
                skipped = "c" in skip
            else:
                skipped = "gaussians" in skip or (n == "opacity" and "opacity" in skip)
            if skipped:
                continue
            steps[n] += 1
            lr = {"xyz": tr.xyz_lr(it), "features": lr_feat, "c": tr.c_lr}.get(n, tr.lrs.get(n))
            eps = 1e-8 if n == "c" else 1e-15
            g_ = grads[n].double()
            m[n] = (0.9 * m[n].double() + 0.1 * g_).float()
            v[n] = (0.999 * v[n].double() + 0.001 * g_ * g_).float()
            bc1, bc2 = 1 - 0.9 ** steps[n], 1 - 0.999 ** steps[n]
            upd = (m[n].double() / bc1) / ((v[n].double() / bc2).sqrt() + eps)
            ref[n].data = (ref[n].data.double() - (lr if torch.is_tensor(lr) else float(lr)) * upd).float()
    torch.cuda.synchronize()
    assert tr.steps == {"gauss": 4, "opacity": 3, "c": 4}
    for n in names:
        a, b = seg(tr.flat, n), ref[n].data
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max())), n
        assert rel_l2(seg(tr.exp_avg, n).cpu().numpy(), m[n].cpu().numpy()) <= 1e-6, n


def _trainer_from_groups(groups, c=0.17):
    from event_3dgs_amd.train_step import EventTrainer
    params = {"xyz": groups["xyz"][0], "features_dc": groups["f_dc"][0], "features_rest": groups["f_rest"][0],
              "opacity": groups["opacity"][0], "scaling": groups["scaling"][0], "rotation": groups["rotation"][0]}
    tr = EventTrainer({k: v.to(DEV) for k, v in params.items()}, DEV, c_init=c)
    tr._build({k: [t.to(DEV) for t in v] for k, v in groups.items()}, c_value=c, c_moments=(0.25, 0.5))
    return tr


def test_g8_densification_golden_replayed_on_the_device():
    """Golden G8 (the reference's own GaussianModel.densify_and_prune, tests/golden/make_golden.py) through the HIP
    plan / apply kernels: same rows, same order, same parameters, Adam moments and statistics."""
    from event_3dgs_amd import densify
    names = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
    g = np.load(os.path.join(GOLDEN, "densify.npz"))
    groups = {n: [torch.tensor(g[f"in_{n}"]), torch.tensor(g[f"in_{n}_m"]), torch.tensor(g[f"in_{n}_v"])] for n in names}
    tr = _trainer_from_groups(groups)
    stats = densify.DensifyStats(256, DEV)
    stats.xyz_gradient_accum = torch.tensor(g["in_accum"]).to(DEV)
    stats.denom = torch.tensor(g["in_denom"]).to(DEV)
    stats.max_radii2D = torch.tensor(g["in_maxr"]).to(DEV)
    max_grad, min_opacity, extent, max_screen, pdense = (float(v) for v in g["args"])

    def cpu_sampler(stds):                       # the golden run drew the split offsets from torch's CPU generator
        torch.manual_seed(77)
        stds = stds.cpu()
        return torch.normal(mean=torch.zeros((stds.size(0), 3)), std=stds)
    n = tr.densify_and_prune(stats, max_grad, min_opacity, extent, max_screen, pdense, sampler=cpu_sampler)
    assert n == g["out_xyz"].shape[0] and n != 256
    out = tr.export_groups()
    for name in names:
        for j, suf in enumerate(("", "_m", "_v")):
            ref = g[f"out_{name}{suf}"]
            got = out[name][j].cpu().numpy()
            assert got.shape == ref.shape, (name, suf)
            assert np.allclose(got, ref, rtol=2e-6, atol=1e-7), (name, suf, np.abs(got - ref).max())
    assert np.array_equal(stats.xyz_gradient_accum.cpu().numpy(), g["out_accum"])
    assert np.array_equal(stats.denom.cpu().numpy(), g["out_denom"])
    assert np.array_equal(stats.max_radii2D.cpu().numpy(), g["out_maxr"])
    off, _ = tr.seg["c"]
    assert abs(float(tr.c) - 0.17) < 1e-7 and float(tr.exp_avg[off]) == 0.25 and float(tr.exp_avg_sq[off]) == 0.5
    tr.reset_opacity()
    out = tr.export_groups()
    assert np.allclose(out["opacity"][0].cpu().numpy(), g["reset_opacity"], rtol=2e-6, atol=1e-7)
    assert float(out["opacity"][1].abs().sum()) == 0.0
    assert np.allclose(out["xyz"][1].cpu().numpy(), g["reset_xyz_m"])


@pytest.mark.parametrize("size_prune", [None, 20])
def test_device_densification_equals_the_torch_form(size_prune):
    """50 k Gaussians with trained-looking statistics: the kernels against densify.densify_and_prune (the golden-pinned
    torch form) on the exported groups, fed the same split noise."""
    from event_3dgs_amd import densify
    from event_3dgs_amd.train_step import EventTrainer
    params, _ = _scene(N=50_000)
    tr = EventTrainer(params, DEV)
    g = torch.Generator().manual_seed(4)
    tr.exp_avg.copy_(torch.randn(tr.flat.shape, generator=g)); tr.exp_avg_sq.copy_(torch.rand(tr.flat.shape, generator=g))
    N = tr.N
    mk = lambda: densify.DensifyStats(N, DEV)
    sa, sb = mk(), mk()
    den = torch.randint(0, 5, (N, 1), generator=g).float()
    acc = torch.rand(N, 1, generator=g) * 6e-4 * den
    for s in (sa, sb):
        s.xyz_gradient_accum = acc.to(DEV).clone(); s.denom = den.to(DEV).clone()
    extent = 2.0                                  # percent_dense * extent = 0.02: about half of the selected rows split
    groups = tr.export_groups()
    noise = {}

    def sampler(stds):
        noise["z"] = torch.normal(mean=torch.zeros((stds.size(0), 3), device=DEV), std=stds)
        return noise["z"]
    n_dev = tr.densify_and_prune(sa, 0.0002, 0.005, extent, size_prune, 0.01, sampler=sampler)
    real_normal = torch.normal
    try:
        torch.normal = lambda mean, std: noise["z"]          # the torch form draws inside; hand it the same numbers
        n_ref = densify.densify_and_prune(groups, sb, 0.0002, 0.005, extent, size_prune, 0.01)
    finally:
        torch.normal = real_normal
    assert n_dev == n_ref and n_dev != N and noise["z"].shape[0] > 100
    out = tr.export_groups()
    for name in groups:
        for j in range(3):
            a, b = out[name][j], groups[name][j]
            assert a.shape == b.shape, name
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), (name, j, float((a - b).abs().max()))
    assert torch.equal(sa.denom, sb.denom) and sa.max_radii2D.shape == (n_dev,)
    # and the trainer keeps training on the new buffers
    cams = _scene(N=10)[1]
    bg = torch.zeros(3, device=DEV)
    gts = [tr.render_raw(c, bg)["color"].clone() for c in cams]
    sc = tr.step(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg)
    assert torch.isfinite(sc).all() and torch.isfinite(tr.flat).all()


def _kernel_activations(leaves):
    """exp / normalize / sigmoid of the raw parameters with the VALUES the E3DGS_FLAG_PREACT kernels compute (the oracle's
    gso_activate, scene/gaussian_model.py:33-41) and torch's autograd chain rule (straight-through on the <= 2e-6
    relative difference between exp_det and torch.exp)."""
    from oracle import c_oracle
    dev = leaves["scaling"].device
    s, q, o = (torch.from_numpy(a).to(dev) for a in c_oracle.activate(
        leaves["scaling"].detach().cpu().numpy(), leaves["rotation"].detach().cpu().numpy(),
        leaves["opacity"].detach().cpu().numpy()))
    ts, tq, to = torch.exp(leaves["scaling"]), torch.nn.functional.normalize(leaves["rotation"]), torch.sigmoid(leaves["opacity"])
    return ts + (s - ts).detach(), tq + (q - tq).detach(), to + (o.reshape(to.shape) - to).detach()


def test_triplet_with_mixed_resolutions_runs_as_two_multi_view_passes():
    """utils/camera_utils.py:19-52 sizes every image on its own: an intensity frame at another resolution than the event
    pair.  The trainer renders [intensity] and [now, next] as two multi-view passes (EventTrainer._compute_gradients_two_sizes);
    gradients equal the autograd composition of the reference's formulas (torch) over the drop-in operator, and the whole
    step -- SH gradient rebuilt from the three views' colour gradients inside the SH optimizer kernel -- equals
    compute_gradients() + apply_update() with the SH gradient in memory."""
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.rasterizer import rasterize_gaussians
    from event_3dgs_amd.train_step import EventTrainer
    from oracle import torch_oracle
    params, _ = _scene(N=2500)
    cams = [orbit_camera(0, 16, 176, 128, device=DEV), orbit_camera(0, 16, 128, 96, device=DEV, daz=0.004),
            orbit_camera(0, 16, 128, 96, device=DEV, daz=0.012)]
    bg = torch.ones(3, device=DEV)
    gts = _gts(params, cams, bg)
    gts[2][:, :, :48] = gts[1][:, :, :48]          # no event on a third of the frame: rho < 1, the intensity term is live
    for blur in (None, (0.9 * gts[0]).contiguous()):
        tr = EventTrainer(params, DEV, track_densification_stats=True)
        sc = tr.compute_gradients(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg, gt_blur=blur)
        leaves = {k: t.detach().clone().requires_grad_(True) for k, t in tr.views.items()}
        c = tr.c.detach().clone().requires_grad_(True)
        feats = leaves["features"].t().reshape(tr.N, 16, 3)
        # the VALUES of the kernels' deterministic activations (gso_activate) with torch's chain rule: the L1 terms make the
        # gradient discontinuous in the image (sign(e)), so the two sides must render the same bits to be comparable
        act_s, act_q, act_o = _kernel_activations(leaves)
        imgs = []
        for cam in cams:
            m2 = torch.zeros_like(leaves["xyz"], requires_grad=True)
            img, _ = rasterize_gaussians(leaves["xyz"], m2, feats, None, act_o, act_s, act_q, None, tr._settings(cam, bg))
            imgs.append(img)
        ref = torch_oracle.event_iteration_loss(imgs[0], imgs[1], imgs[2], gts[0], gts[1], gts[2], c, gt_blur=blur)
        ref.backward()
        assert abs(float(sc[0]) - float(ref)) <= 2e-5 * abs(float(ref))
        assert abs(float(tr.c_grad) - float(c.grad)) <= 2e-4 * abs(float(c.grad))
        for name in ("xyz", "features", "opacity", "scaling", "rotation"):
            a, b = tr.grads[name].cpu().numpy(), leaves[name].grad.cpu().numpy()
            assert rel_l2(a, b) <= 1e-3, (name, rel_l2(a, b))
        assert float(tr.viewspace_grad.abs().max()) > 0
        tr.apply_update()
        assert torch.isfinite(tr.flat).all()
        # step() (colour-gradient route + deferred SH position term) against the form with the whole gradient in memory: the
        # two passes' position gradients are added before / after the SH view-direction terms -- fp32 association, not bits
        a, b = EventTrainer(params, DEV), EventTrainer(params, DEV)
        for it in range(3):
            sa = a.step(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg, gt_blur=blur)
            sb = b.compute_gradients(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg, gt_blur=blur).clone()
            b.apply_update()
            if it == 0:
                assert torch.equal(sa[:5], sb[:5])
            assert torch.allclose(sa[:5], sb[:5], rtol=1e-4, atol=0)
        assert rel_l2(a.exp_avg.cpu().numpy(), b.exp_avg.cpu().numpy()) <= 1e-5
        assert rel_l2(a.exp_avg_sq.cpu().numpy(), b.exp_avg_sq.cpu().numpy()) <= 1e-5
        f_off, f_n = a.seg["features"]
        assert torch.equal(a.exp_avg[f_off:f_off + f_n], b.exp_avg[f_off:f_off + f_n]) or \
            rel_l2(a.exp_avg[f_off:f_off + f_n].cpu().numpy(), b.exp_avg[f_off:f_off + f_n].cpu().numpy()) <= 1e-5
        assert a.count_retries == 0 and a._packed_views == 0


@pytest.mark.parametrize("deblur", [False, True])
def test_sh_optimizer_from_colour_gradients_is_bit_identical(deblur):
    """step() on one rank: backward hands out per-view colour gradients, e3dgs_sh_adam_from_colour rebuilds the SH gradient
    in registers and applies Adam -- against compute_gradients() + apply_update() with the SH gradient in memory:
    parameters, both moments and c bit for bit over several iterations, also with the opacity group's step count lagging
    and the SH degree below 3 (inactive coefficients still decay)."""
    from event_3dgs_amd.train_step import EventTrainer
    params, cams = _scene(N=3000)
    bg = torch.zeros(3, device=DEV)
    gts = [(torch.round(t.clamp(0, 1) * 255) / 255).contiguous() for t in _gts(params, cams, bg)]
    blur = (0.5 * (gts[0] + gts[2])).contiguous() if deblur else None
    a = EventTrainer(params, DEV, track_densification_stats=True, active_sh_degree=2)
    b = EventTrainer(params, DEV, track_densification_stats=True, active_sh_degree=2)
    assert a.sh_via_colour
    for it in range(5):
        sa = a.step(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg, gt_blur=blur)
        sb = b.compute_gradients(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg, gt_blur=blur).clone()
        assert b._packed_views == 0
        b.apply_update()
        assert torch.equal(sa, sb), it
        assert torch.equal(a.flat, b.flat) and torch.equal(a.exp_avg, b.exp_avg) and torch.equal(a.exp_avg_sq, b.exp_avg_sq), it
        assert torch.equal(a.viewspace_grad, b.viewspace_grad)
        if it == 1:                      # make the opacity group's step count lag behind the others
            for t in (a, b):
                t.reset_opacity()
                t.compute_gradients(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg,
                                    sh_via_colour=t is a)
                t.apply_update(skip=("opacity",))
            assert a.steps == b.steps == {"gauss": 3, "opacity": 2, "c": 3}
            assert torch.equal(a.flat, b.flat)
            a.active_sh_degree = b.active_sh_degree = 3
    assert a.steps == {"gauss": 6, "opacity": 5, "c": 6} and a.iteration == 6


@pytest.mark.parametrize("mode", ["event", "gray"])
def test_iteration_without_host_wait_is_bit_identical_and_survives_an_overflow(mode, monkeypatch):
    """EventTrainer sizes the binning buffers from earlier counts and reads the count on the device
    (e3dgs_rasterize_forward_multi_capacity): same parameters, bit for bit, as begin -> host wait -> finish; a capacity
    that turns out too small costs a repeated forward / backward (count_retries), never a wrong update."""
    from event_3dgs_amd.train_step import EventTrainer
    params, cams = _scene(N=6000)
    bg = torch.zeros(3, device=DEV)
    gts = _gts(params, cams, bg)

    def run(no_wait, sabotage=False):
        monkeypatch.setenv("E3DGS_NO_HOST_WAIT", "1" if no_wait else "0")
        tr = EventTrainer(params, DEV)
        assert tr.no_host_wait == no_wait
        for it in range(4):
            if sabotage and it == 2:
                (key, cap), = tr._capacity.items()
                tr._capacity[key] = 1000                 # far below the real count: the forward emits nothing
            if mode == "event":
                tr.step(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg)
            else:
                tr.step_image(cams[0], gts[0], bg, mode="gray")
        torch.cuda.synchronize()
        return tr
    ref = run(False)
    new = run(True)
    assert new.count_retries == 0 and len(new._capacity) == 1
    assert torch.equal(ref.flat, new.flat) and torch.equal(ref.exp_avg_sq, new.exp_avg_sq)
    hit = run(True, sabotage=True)
    assert hit.count_retries == 1
    assert torch.equal(ref.flat, hit.flat) and torch.equal(ref.exp_avg_sq, hit.exp_avg_sq)
    (key, cap), = hit._capacity.items()
    assert cap > 1000                                # regrown from the count that did not fit


def test_checkpoint_resume_continues_the_adam_bias_correction():
    """capture_checkpoint(EventTrainer.steps) -> restore_checkpoint / restored_steps -> import_groups(steps=...): the
    resumed trainer takes exactly the step the original takes next (a resume that restarted the step counts at 1 with
    non-zero moments would move the parameters ~10x too far)."""
    from event_3dgs_amd import io_formats as IO
    from event_3dgs_amd.densify import DensifyStats
    from event_3dgs_amd.train_step import EventTrainer
    params, cams = _scene()
    bg = torch.zeros(3, device=DEV)
    gts = _gts(params, cams, bg)
    a = EventTrainer(params, DEV)
    for _ in range(3):
        a.step(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg)
    a.reset_opacity()                                     # the opacity group's moments restart; its count does not
    a.steps["opacity"] -= 1                               # (as after an iteration torch skipped for the opacity group)
    lrs = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 2.5e-3 / 20, "opacity": 0.05, "scaling": 5e-3, "rotation": 1e-3}
    tup = IO.capture_checkpoint(a.export_groups(), DensifyStats(a.N, DEV), a.active_sh_degree, 1.0, lrs, a.steps)
    groups, _, deg, _ = IO.restore_checkpoint(tup)
    b = EventTrainer(params, DEV, active_sh_degree=deg)
    b.import_groups(groups, steps={**IO.restored_steps(tup), "c": a.steps["c"]})
    off, _ = a.seg["c"]
    for dst, src in ((b.flat, a.flat), (b.exp_avg, a.exp_avg), (b.exp_avg_sq, a.exp_avg_sq)):
        dst[off] = src[off]                               # optimizer_c is not part of GaussianModel.capture()
    b.iteration = a.iteration
    assert b.steps == a.steps and torch.equal(a.flat, b.flat) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)
    for t in (a, b):
        t.step(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg)
    torch.cuda.synchronize()
    assert torch.equal(a.flat, b.flat)


def test_shared_pose_iteration_equals_three_renders():
    """The reference's datasets give the event camera `index` the pose of the training camera `index`
    (scene/dataset_readers.py:157 reads both with the same extrinsics), so renders #1 and #2 of an event iteration
    (train.py:144,159) are the same render.  EventTrainer renders it once and sends the sum of the two pixel gradients
    through its backward: same loss bits, gradients and update equal to summation order; not taken when the densification
    statistics need the screen-space gradient of render #1 alone."""
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer
    params, cams = _scene()
    W, H = cams[0].image_width, cams[0].image_height
    same = orbit_camera(0, 16, W, H, device=DEV, daz=0.0)            # another camera object, the pose of cams[0]
    assert same is not cams[0]
    bg = torch.tensor([0.2, 0.2, 0.2], device=DEV)
    gts = _gts(params, cams, bg)
    blur = (0.5 * (gts[0] + gts[2])).contiguous()
    for gt_blur in (None, blur):
        a, b = EventTrainer(params, DEV), EventTrainer(params, DEV)
        b.share_coincident_views = False
        sa = a.compute_gradients(cams[0], same, cams[2], gts[0], gts[1], gts[2], bg, gt_blur=gt_blur).clone()
        sb = b.compute_gradients(cams[0], same, cams[2], gts[0], gts[1], gts[2], bg, gt_blur=gt_blur).clone()
        torch.cuda.synchronize()
        assert a.shared_pose_iterations == 1 and b.shared_pose_iterations == 0
        assert torch.equal(sa[:6], sb[:6])                              # loss, dL/dc, rho, the three L1 terms: same bits
        assert float(a.c_grad) == float(b.c_grad)
        for name in ("xyz", "features", "opacity", "scaling", "rotation"):
            ga, gb = a.grads[name].cpu().numpy(), b.grads[name].cpu().numpy()
            assert np.abs(gb).max() > 0
            assert rel_l2(ga, gb) <= 1e-5, (name, rel_l2(ga, gb))
        # the whole step, as training runs it (SH gradient rebuilt from TWO views' colour gradients)
        a2, b2 = EventTrainer(params, DEV), EventTrainer(params, DEV)
        b2.share_coincident_views = False
        for _ in range(3):
            a2.step(cams[0], same, cams[2], gts[0], gts[1], gts[2], bg, gt_blur=gt_blur)
            b2.step(cams[0], same, cams[2], gts[0], gts[1], gts[2], bg, gt_blur=gt_blur)
        torch.cuda.synchronize()
        assert a2.shared_pose_iterations == 3
        assert rel_l2(a2.exp_avg.cpu().numpy(), b2.exp_avg.cpu().numpy()) <= 1e-4
        assert float((a2.flat - b2.flat).abs().max()) <= 0.05
    # distinct poses: three renders
    c = EventTrainer(params, DEV)
    c.compute_gradients(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg)
    assert c.shared_pose_iterations == 0
    # a camera moved in place afterwards (pose refinement) is noticed: the cached answer is tied to the tensors' versions
    e = EventTrainer(params, DEV)
    moved = orbit_camera(0, 16, W, H, device=DEV, daz=0.0)
    e.compute_gradients(cams[0], moved, cams[2], gts[0], gts[1], gts[2], bg)
    assert e.shared_pose_iterations == 1
    other = orbit_camera(0, 16, W, H, device=DEV, daz=0.004)
    for name in ("world_view_transform", "full_proj_transform", "camera_center"):
        getattr(moved, name).copy_(getattr(other, name))
    e.compute_gradients(cams[0], moved, cams[2], gts[0], gts[1], gts[2], bg)
    assert e.shared_pose_iterations == 1
    f3 = EventTrainer(params, DEV)
    f3.compute_gradients(cams[0], other, cams[2], gts[0], gts[1], gts[2], bg)
    torch.cuda.synchronize()
    for name in ("xyz", "features", "opacity"):
        assert torch.equal(e.grads[name], f3.grads[name]), name          # really rendered at the new pose
    # densification statistics wanted: render #1's OWN screen-space gradient (train.py:145,317-320) -- the shared view's
    # tiles run a second dL/dalpha chain on render #1's pixel gradient (e3dgs_rasterize_backward_multi_stats): same
    # statistics input as with three renders, and the same parameter gradients as without statistics
    for gt_blur in (None, blur):
        d = EventTrainer(params, DEV, track_densification_stats=True)
        d.SHARE_STATS_MIN_INSTANCES = d.SHARE_STATS_MIN_TILES = 0      # (by default only launches that fill the GPU take this path)
        d3 = EventTrainer(params, DEV, track_densification_stats=True)
        d3.share_coincident_views = False
        sd = d.compute_gradients(cams[0], same, cams[2], gts[0], gts[1], gts[2], bg, gt_blur=gt_blur).clone()
        s3 = d3.compute_gradients(cams[0], same, cams[2], gts[0], gts[1], gts[2], bg, gt_blur=gt_blur).clone()
        torch.cuda.synchronize()
        assert d.shared_pose_iterations == 1 and d3.shared_pose_iterations == 0
        assert torch.equal(sd[:6], s3[:6])
        va, vb = d.viewspace_grad.cpu().numpy(), d3.viewspace_grad.cpu().numpy()
        assert np.abs(vb).max() > 0 and np.all(va[:, 2] == 0)
        assert rel_l2(va, vb) <= 1e-5, rel_l2(va, vb)
        # ... and it is NOT the gradient of the summed pixel gradients
        n = EventTrainer(params, DEV)
        n.compute_gradients(cams[0], same, cams[2], gts[0], gts[1], gts[2], bg, gt_blur=gt_blur)
        for name in ("xyz", "features", "opacity", "scaling", "rotation"):
            assert torch.equal(d.grads[name], n.grads[name]), name          # the first chain is untouched by the second
            assert rel_l2(d.grads[name].cpu().numpy(), d3.grads[name].cpu().numpy()) <= 1e-5, name
    d.compute_gradients(cams[0], same, cams[2], gts[0], gts[1], gts[2], bg, viewspace_grad=False)
    assert d.shared_pose_iterations == 2
    # default threshold: a launch this small keeps three renders on statistics iterations
    small = EventTrainer(params, DEV, track_densification_stats=True)
    small.compute_gradients(cams[0], same, cams[2], gts[0], gts[1], gts[2], bg)
    small.compute_gradients(cams[0], same, cams[2], gts[0], gts[1], gts[2], bg)
    assert small.shared_pose_iterations == 0 and 0 < small._instances_per_view < small.SHARE_STATS_MIN_INSTANCES


def test_scene_directory_to_shared_pose_training(tmp_path):
    """A scene directory in the reference's layout (sparse/0 + images + images_event: the golden COLMAP model with random
    frames) -> load_colmap_scene -> create_from_pcd -> fit_event_scene.  The loader gives event camera `index` the pose
    of training camera `index` (as scene/dataset_readers.py:157 does), so every event iteration renders two views instead
    of three -- with the statistics chain while the densification statistics are collected."""
    import shutil
    from PIL import Image
    from event_3dgs_amd import fit, scene_io
    from simple_knn._C import distCUDA2
    cd = os.path.join(GOLDEN, "colmap_tiny")
    root = str(tmp_path / "scene")
    os.makedirs(os.path.join(root, "sparse/0"))
    for f in ("cameras.bin", "images.bin", "points3D.bin"):
        shutil.copy(os.path.join(cd, f), os.path.join(root, "sparse/0", f))
    from event_3dgs_amd import io_formats as IO
    rs = np.random.RandomState(0)
    for d in ("images", "images_event", "renders"):
        os.makedirs(os.path.join(root, d))
        for im in IO.read_images_binary(os.path.join(cd, "images.bin")).values():
            Image.fromarray(rs.randint(0, 256, (48, 64, 3), dtype=np.uint8)).save(os.path.join(root, d, im.name))
    sc = scene_io.load_colmap_scene(root, gray=True, event=True, device=DEV)
    params = scene_io.create_from_pcd(sc.point_cloud, sc.cameras_extent, distCUDA2, device=DEV)
    from event_3dgs_amd.train_step import EventTrainer
    for threshold, expect in ((None, [0, 0, 0, 0, 0, 1, 2, 3, 4, 5]), (0, list(range(1, 11)))):
        # iterations 1..5 collect statistics: three renders for a launch this small (default threshold), or -- threshold
        # 0 -- the shared view with the second gradient chain; 6..10 share the pose in both runs
        shared_after = []
        keep = (EventTrainer.SHARE_STATS_MIN_INSTANCES, EventTrainer.SHARE_STATS_MIN_TILES)
        if threshold is not None:
            EventTrainer.SHARE_STATS_MIN_INSTANCES = EventTrainer.SHARE_STATS_MIN_TILES = threshold
        try:
            tr = fit.fit_event_scene(params, sc.train_cameras, sc.event_cameras, torch.zeros(3, device=DEV), DEV,
                                     iterations=10, cameras_extent=sc.cameras_extent, densify_until_iter=6,
                                     densify_from_iter=2, densification_interval=2, rng=lambda a, b: 2,
                                     on_iteration=lambda it, t, s: shared_after.append(t.shared_pose_iterations))
        finally:
            EventTrainer.SHARE_STATS_MIN_INSTANCES, EventTrainer.SHARE_STATS_MIN_TILES = keep
        torch.cuda.synchronize()
        assert shared_after == expect, (threshold, shared_after)
    assert torch.isfinite(tr.flat).all()
    # the one-render modes and the evaluation protocol on the same loaded cameras (frames with PILtoTorch's strides
    # are laid out as planes on the way in)
    strided = sc.train_cameras[2].original_image.permute(1, 2, 0).contiguous().permute(2, 0, 1)
    assert not strided.is_contiguous()
    sc.train_cameras[2].original_image = strided
    sc.event_cameras[2].original_image = sc.event_cameras[2].original_image.permute(1, 2, 0).contiguous().permute(2, 0, 1)
    for mode in ("gray", "rgb", "event"):
        t2 = fit.fit_event_scene(params, sc.train_cameras, sc.event_cameras, torch.zeros(3, device=DEV), DEV, iterations=3,
                                 cameras_extent=sc.cameras_extent, rng=lambda a, b: 2, mode=mode)
        assert torch.isfinite(t2.flat).all(), mode
    res = scene_io.evaluate_views(lambda cam: tr.render_raw(cam, torch.zeros(3, device=DEV))["color"], sc.test_cameras,
                                  index_list=(1, 3))
    assert np.isfinite(res["psnr"]) and np.isfinite(res["ssim"])


def test_event_loss_kernel_shared_render_conventions():
    """e3dgs_event_loss when render #1 and render #2 are one image: aliased outputs -> the sum; separate outputs -> d_now =
    the sum, d_image = the intensity term alone; the autograd wrapper still hands autograd two separate gradients."""
    from event_3dgs_amd import losses
    g = torch.Generator(device=DEV).manual_seed(3)
    H, W = 37, 53
    img, nxt, gi, gn, gx = (torch.rand(3, H, W, device=DEV, generator=g) * 0.8 + 0.1 for _ in range(5))
    c = torch.tensor([0.21], device=DEV)
    sc0, di0, dn0, dx0 = (t.clone() for t in losses.event_loss_raw(img, img.clone(), nxt, c, gi, gn, gx))
    buf = lambda: torch.empty_like(img)
    sh = buf()
    sc1, _, _, dx1 = losses.event_loss_raw(img, img, nxt, c, gi, gn, gx,
                                          out=(torch.empty(8, device=DEV), sh, sh, buf(), torch.empty(
                                              _lib_scratch(W, H), dtype=torch.uint8, device=DEV)))
    assert torch.equal(sc1[:6], sc0[:6]) and torch.equal(dx1, dx0) and torch.equal(sh, di0 + dn0)
    sc2, di2, dn2, dx2 = losses.event_loss_raw(img, img, nxt, c, gi, gn, gx)
    assert torch.equal(di2, di0) and torch.equal(dn2, di0 + dn0) and torch.equal(dx2, dx0)
    leaf = img.clone().requires_grad_(True)
    losses.event_iteration_loss(leaf, leaf, nxt, c, gi, gn, gx).backward()
    assert torch.allclose(leaf.grad, di0 + dn0, rtol=0, atol=0)


def test_event_loss_one_sweep_for_a_known_ground_truth_pair():
    """e3dgs_event_loss_cached: the first call on a ground-truth pair runs the three launches and leaves the pair's
    count(D* != 0); later calls sweep the images ONCE (event_fused_kernel) -- scalars and the three pixel gradients bit for
    bit those of e3dgs_event_loss, for odd frame sizes (scalar path), 16-byte friendly ones (vector path), the deblur term
    and both shared-render conventions; a frame changed in place starts over."""
    from event_3dgs_amd import losses
    g = torch.Generator(device=DEV).manual_seed(11)
    for H, W, blur in ((37, 53, False), (64, 96, False), (64, 96, True)):
        img, now, nxt, gi, gn, gx, gb = (torch.rand(3, H, W, device=DEV, generator=g) * 0.8 + 0.1 for _ in range(7))
        gx[:, : H // 3] = gn[:, : H // 3]                       # a third of the contrast targets exactly zero: rho < 1
        c = torch.tensor([0.19], device=DEV)
        blur_t = gb if blur else None
        ref = [t.clone() for t in losses.event_loss_raw(img, now, nxt, c, gi, gn, gx, blur_t)]
        pc = losses.PairCounts()
        first = [t.clone() for t in losses.event_loss_raw(img, now, nxt, c, gi, gn, gx, blur_t, pair_counts=pc)]
        cnt = pc.lookup(gn, gx, 0.17)
        assert cnt is not None and float(cnt) == H * W - (H // 3) * W   # (random targets elsewhere: non-zero)
        dc = torch.zeros(1, device=DEV)
        second = [t.clone() for t in losses.event_loss_raw(img, now, nxt, c, gi, gn, gx, blur_t, pair_counts=pc, dc_out=dc)]
        for a, b, e in zip(ref, first, second):
            assert torch.equal(a, b) and torch.equal(a, e)
        assert float(dc) == float(ref[0][1])
        # other renders, same pair: still one sweep, still identical to the three-launch form
        img2, now2, nxt2 = (torch.rand(3, H, W, device=DEV, generator=g) * 0.8 + 0.1 for _ in range(3))
        ref2 = [t.clone() for t in losses.event_loss_raw(img2, now2, nxt2, c, gi, gn, gx, blur_t)]
        got2 = losses.event_loss_raw(img2, now2, nxt2, c, gi, gn, gx, blur_t, pair_counts=pc)
        assert all(torch.equal(a, b) for a, b in zip(ref2, got2))
        # shared render: aliased outputs (the sum) and separate outputs (sum + intensity part)
        sh_ref = [t.clone() for t in losses.event_loss_raw(img2, img2, nxt2, c, gi, gn, gx, blur_t)]
        sh_got = losses.event_loss_raw(img2, img2, nxt2, c, gi, gn, gx, blur_t, pair_counts=pc)
        assert all(torch.equal(a, b) for a, b in zip(sh_ref, sh_got))
        buf = torch.empty_like(img2)
        scr = torch.empty(_lib_scratch(W, H), dtype=torch.uint8, device=DEV)
        _, _, _, dxa = losses.event_loss_raw(img2, img2, nxt2, c, gi, gn, gx, blur_t, pair_counts=pc,
                                             out=(torch.empty(8, device=DEV), buf, buf, torch.empty_like(img2), scr))
        assert torch.equal(buf, sh_ref[2])                              # (separate outputs: d_now holds the same sum)
        assert torch.equal(dxa, sh_ref[3])
        # a ground-truth frame modified in place is a new pair
        gx.mul_(0.5)
        assert pc.lookup(gn, gx, 0.17) is None
        ref3 = [t.clone() for t in losses.event_loss_raw(img, now, nxt, c, gi, gn, gx, blur_t)]
        got3 = [t.clone() for t in losses.event_loss_raw(img, now, nxt, c, gi, gn, gx, blur_t, pair_counts=pc)]
        got4 = losses.event_loss_raw(img, now, nxt, c, gi, gn, gx, blur_t, pair_counts=pc)
        assert all(torch.equal(a, b) and torch.equal(a, e) for a, b, e in zip(ref3, got3, got4))


def _lib_scratch(W, H):
    from event_3dgs_amd import _lib
    return _lib.lib().e3dgs_event_loss_scratch_bytes(W, H)


def test_two_trainers_with_different_options_on_concurrent_streams():
    """The library reads no mutable process-wide state once a call carries E3DGS_FLAG_OPTIONS: two trainers with DIFFERENT
    rasteriser options (reference rectangle binning + large-scene decomposition vs. exact tile culling + small-scene
    decomposition) run from two host threads on two streams at the same time and produce exactly what each produces
    alone (instance counts, images, gradients bit for bit -- the backward has no atomics)."""
    import threading
    from event_3dgs_amd.train_step import EventTrainer
    params, cams = _scene(N=4000, W=208, H=160)
    bg = torch.zeros(3, device=DEV)
    gts = _gts(params, cams, bg)
    opts = (dict(tile_cull=0, small_scene_paths=False), dict(tile_cull=1, small_scene_paths=True))

    def run(opt, stream, out, reps=6):
        tr = EventTrainer(params, DEV, **opt)
        tr.no_host_wait = False
        res = []
        with torch.cuda.stream(stream):
            for _ in range(reps):
                tr.flat_grad.fill_(float("nan"))
                sc = tr.compute_gradients(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg)
                imgs = tr._pool.typed("out_color", (3, 3, cams[0].image_height, cams[0].image_width))
                res.append((tr.flat_grad.clone(), imgs.clone(), sc.clone()))
            stream.synchronize()
        out.append((res, tr))

    serial = []
    for opt in opts:
        run(opt, torch.cuda.Stream(DEV), serial)
    both = [[], []]
    threads = [threading.Thread(target=run, args=(opt, torch.cuda.Stream(DEV), both[k])) for k, opt in enumerate(opts)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    for k in range(2):
        (res_c, _), (res_s, _) = both[k][0], serial[k]
        for (g_c, im_c, sc_c), (g_s, im_s, sc_s) in zip(res_c, res_s):
            assert torch.isfinite(g_c).all()
            assert torch.equal(g_c, g_s) and torch.equal(im_c, im_s) and torch.equal(sc_c, sc_s)
    # the options were honoured: the images agree (culling is invisible), the gradients only to summation order
    assert torch.equal(serial[0][0][0][1], serial[1][0][0][1])
    a, b = serial[0][0][0][0], serial[1][0][0][0]
    assert not torch.equal(a, b) and rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-4
    # ... and the instance counts differ (rectangle binning keeps what exact culling drops)
    from event_3dgs_amd import rasterizer
    counts = []
    for opt in opts:
        tr = EventTrainer(params, DEV, **opt)
        v = tr.views
        raw = rasterizer.forward_multi(v["xyz"], v["features"], v["opacity"], v["scaling"], v["rotation"],
                                       [tr._settings(c, bg) for c in cams], flags=tr.FWD_FLAGS)
        counts.append(raw["num_rendered"])
    assert counts[0] > counts[1] > 0
