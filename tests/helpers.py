"""Shared scene builders for the parity tests (seeded, small enough for the CPU oracle)."""
import math

import numpy as np
import torch

from event_3dgs_amd import synth
from event_3dgs_amd.cameras import orbit_camera


def scene(N, W, H, seed=0, kind="trained", k=0, K=8, scale_boost=1.0, radius=4.0):
    params = synth.make_scene(N, kind, seed=seed)
    act = synth.activate(params)
    act["scales"] = act["scales"] * scale_boost
    cam = orbit_camera(k, K, W, H, radius=radius)
    g = torch.Generator().manual_seed(seed + 100)
    act["colors"] = torch.rand(N, 3, generator=g)
    return act, cam


def oracle_kwargs(act, cam, bg, use_sh, use_cov, sh_degree=3, scale_modifier=1.0):
    from oracle import torch_oracle
    kw = dict(means3D=act["means3D"].numpy(), opacities=act["opacities"].numpy(),
              viewmatrix=cam.world_view_transform.contiguous().numpy(), projmatrix=cam.full_proj_transform.numpy(),
              campos=cam.camera_center.contiguous().numpy(), bg=np.asarray(bg, np.float32), width=cam.image_width,
              height=cam.image_height, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
              scale_modifier=scale_modifier)
    if use_sh:
        kw.update(shs=act["shs"].numpy(), sh_degree=sh_degree)
    else:
        kw.update(colors_precomp=act["colors"].numpy())
    if use_cov:
        kw.update(cov3D_precomp=torch_oracle.build_cov3d(act["scales"], act["rotations"], scale_modifier).numpy())
    else:
        kw.update(scales=act["scales"].numpy(), rotations=act["rotations"].numpy())
    return kw


from oracle.metrics import per_gaussian_err, rel_l2  # noqa: E402,F401


def window_parity(trainer, cam, bg, tile_rows, grad_seed=7):
    """Full-size parity on a tile-row window (GPU tests; the C oracle composites only rows [r0, r1) of the frame, which
    it finishes in seconds at 1-2 M Gaussians / 1080p, while projection, culls, radii and binning cover ALL Gaussians).

    Runs the HIP operator and the C oracle on identical activated inputs (the trainer's parameters, activated with torch
    on the CPU) and returns a dict:
      radii_equal / image_max_abs / final_T_equal / n_contrib_equal   -- forward on the window (bit-exact expected)
      grad[name] = (global rel. L2, per-Gaussian max |d_i| / (|ref_i| + 1e-3 max|ref|))   -- backward of a random pixel
                   gradient that is non-zero on the window rows only, so the oracle's windowed walk IS the full gradient
    """
    from event_3dgs_amd import rasterizer
    from oracle import c_oracle
    dev = trainer.device
    v = {k: t.detach().cpu() for k, t in trainer.views.items()}
    means = v["xyz"].numpy()
    scales = torch.exp(v["scaling"]).numpy()
    rots = torch.nn.functional.normalize(v["rotation"]).numpy()
    opac = torch.sigmoid(v["opacity"]).numpy()
    shs = np.ascontiguousarray(v["features"].t().reshape(-1, 16, 3).numpy())
    P = means.shape[0]
    W, H = int(cam.image_width), int(cam.image_height)
    r0, r1 = tile_rows
    y0, y1 = r0 * 16, min(H, r1 * 16)
    rs = trainer._settings(cam, bg)
    f = c_oracle.Forward(means3D=means, opacities=opac, viewmatrix=cam.world_view_transform.contiguous().cpu().numpy(),
                         projmatrix=cam.full_proj_transform.cpu().numpy(),
                         campos=cam.camera_center.contiguous().cpu().numpy(), bg=bg.cpu().numpy(), width=W, height=H,
                         tanfovx=rs.tanfovx, tanfovy=rs.tanfovy, shs=shs, sh_degree=int(rs.sh_degree), scales=scales,
                         rotations=rots, tile_rows=(r0, r1))
    t = lambda a: torch.from_numpy(a).to(dev)
    hip = rasterizer.forward_raw(t(means), t(shs), None, t(opac), t(scales), t(rots), None, rs)
    st = rasterizer.state_views(hip, P, W, H)
    res = {
        "radii_equal": bool(np.array_equal(hip["radii"].cpu().numpy(), f.radii)),
        "image_max_abs": float(np.abs(hip["color"][:, y0:y1].cpu().numpy() - f.out_color[:, y0:y1]).max()),
        "final_T_equal": bool(np.array_equal(st["final_T"][y0:y1].cpu().numpy(), f.final_T[y0:y1])),
        "n_contrib_equal": bool(np.array_equal(st["n_contrib"][y0:y1].cpu().numpy().astype(np.uint32), f.n_contrib[y0:y1])),
        "visible": int((f.radii > 0).sum()), "oracle_instances": f.num_rendered, "hip_instances": hip["num_rendered"],
    }
    gw = np.zeros((3, H, W), np.float32)
    gw[:, y0:y1] = np.random.default_rng(grad_seed).standard_normal((3, y1 - y0, W)).astype(np.float32)
    gb = f.backward(gw)
    e = lambda *sh: torch.full(sh, float("nan"), dtype=torch.float32, device=dev)
    out = dict(means2D=e(P, 3), opacities=e(P, 1), means3D=e(P, 3), sh=e(P, 16, 3), scales=e(P, 3), rots=e(P, 4))
    rasterizer.backward_raw(hip, t(gw), out)
    pairs = dict(means3D=(out["means3D"], gb["means3D"]), means2D=(out["means2D"], gb["means2D"]),
                 opacities=(out["opacities"], gb["opacities"]), shs=(out["sh"], gb["shs"]),
                 scales=(out["scales"], gb["scales"]), rotations=(out["rots"], gb["rotations"]))
    res["grad"] = {}
    for name, (a, b) in pairs.items():
        a = a.cpu().numpy()
        assert np.isfinite(a).all(), name
        b = np.asarray(b).reshape(a.shape)
        res["grad"][name] = (rel_l2(a, b), per_gaussian_err(a, b))
    f.close()
    return res


def assert_window_parity(res, grad_l2=1e-3, grad_pg=1e-3):
    assert res["radii_equal"], "radii differ from the oracle"
    assert res["image_max_abs"] == 0.0, res["image_max_abs"]            # bit-exact forward (DESIGN: arithmetic contract)
    assert res["final_T_equal"]
    assert res["hip_instances"] <= res["oracle_instances"]                 # exact tile culling only ever drops instances
    if res["hip_instances"] == res["oracle_instances"]:                    # n_contrib counts list POSITIONS: equal lists only
        assert res["n_contrib_equal"]
    for name, (l2, pg) in res["grad"].items():
        assert l2 <= grad_l2, (name, "rel L2", l2)
        assert pg <= grad_pg, (name, "per-Gaussian", pg)
