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
    # n_contrib counts list POSITIONS.  With exact tile culling the HIP lists are ordered sub-sequences of the oracle's
    # (rectangle-binned) lists, so positions are compared after mapping every HIP list position to the position of the
    # same Gaussian in the oracle's list of that tile.
    res["n_contrib_mapped_equal"] = _n_contrib_equal_after_mapping(
        st["point_list"].cpu().numpy().astype(np.int64), st["ranges"].cpu().numpy().astype(np.int64),
        st["n_contrib"].cpu().numpy().astype(np.int64), f.point_list.astype(np.int64), f.ranges.astype(np.int64),
        f.n_contrib.astype(np.int64), W, H, r0, r1)
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


def _n_contrib_equal_after_mapping(pl_h, rg_h, nc_h, pl_o, rg_o, nc_o, W, H, r0, r1):
    gx = (W + 15) // 16
    for ty in range(r0, r1):
        for tx in range(gx):
            t = ty * gx + tx
            lh, lo = pl_h[rg_h[t, 0]:rg_h[t, 1]], pl_o[rg_o[t, 0]:rg_o[t, 1]]
            pos_o = {int(g): i + 1 for i, g in enumerate(lo)}           # a Gaussian occurs once per tile list
            try:
                to_o = np.array([0] + [pos_o[int(g)] for g in lh], np.int64)        # HIP position (1-based) -> oracle position
            except KeyError:
                return False                                             # a HIP instance the oracle does not have
            if len(to_o) > 2 and not np.all(np.diff(to_o[1:]) > 0):
                return False                                             # not an ORDERED sub-sequence
            ys, xs = slice(ty * 16, min(H, ty * 16 + 16)), slice(tx * 16, min(W, tx * 16 + 16))
            if not np.array_equal(to_o[nc_h[ys, xs]], nc_o[ys, xs]):
                return False
    return True


def trainer_path_parity(trainer, cams, bg, rows_per_view, grad_seed=11, rank1_views=()):
    """The path bench.py TIMES against the C oracle at full size: ONE multi-view pass (rasterizer.forward_multi /
    backward_multi) with the trainer's flags -- activations inside the kernels (E3DGS_FLAG_PREACT) on the raw parameters,
    coefficient-major SH (E3DGS_FLAG_SH_PLANAR) -- instead of single-view calls on activated AoS inputs (window_parity).
    The oracle activates the raw parameters with its own statement of scene/gaussian_model.py:33-41 (gso_activate:
    deterministic exp / sigmoid / normalize, the operations the kernels perform) and composites one window of tile rows
    per view; the pixel gradients are supported on those windows; its gradients are summed over the views and taken
    through the activations' chain rule (c_oracle.activate_backward, float64) to the raw parameters.

    Same contract as the operator path: radii, final_T and the image bit for bit, gradients <= 1e-3 per Gaussian.

    rank1_views: views whose pixel gradient is a scalar field times the event loss's luminance weights -- what the trainer
    hands the compositing backward for the contrast renders (e3dgs_rasterize_backward_multi_rank1).  The oracle gets the
    expanded (3, H, W) gradient s * w; the HIP call gets s in plane 0 and NaN in the two planes it must not read."""
    from event_3dgs_amd import rasterizer
    from oracle import c_oracle
    dev = trainer.device
    raw_cpu = {k: t.detach().cpu().numpy() for k, t in trainer.views.items()}
    scales_a, rots_a, opac_a = c_oracle.activate(raw_cpu["scaling"], raw_cpu["rotation"], raw_cpu["opacity"])
    shs_np = np.ascontiguousarray(raw_cpu["features"].T.reshape(-1, 16, 3))
    P = raw_cpu["xyz"].shape[0]
    W, H = int(cams[0].image_width), int(cams[0].image_height)
    settings = [trainer._settings(c, bg) for c in cams]
    v = trainer.views
    hip = rasterizer.forward_multi(v["xyz"], v["features"], v["opacity"], v["scaling"], v["rotation"], settings,
                                   flags=trainer.FWD_FLAGS)
    st = rasterizer.state_views_multi(hip, P, W, H) if hasattr(rasterizer, "state_views_multi") else None
    gw = np.zeros((len(cams), 3, H, W), np.float32)
    gw_hip = np.zeros_like(gw)
    rng = np.random.default_rng(grad_seed)
    res = {"views": []}
    acc = {k: 0.0 for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    for k, (cam, (r0, r1)) in enumerate(zip(cams, rows_per_view)):
        y0, y1 = r0 * 16, min(H, r1 * 16)
        if k in rank1_views:
            sfield = rng.standard_normal((y1 - y0, W)).astype(np.float32)
            for ch, wch in enumerate(rasterizer.LUV_WEIGHTS):
                gw[k, ch, y0:y1] = sfield * np.float32(wch)
            gw_hip[k, 0, y0:y1] = sfield
            gw_hip[k, 1:] = np.nan
        else:
            gw[k, :, y0:y1] = rng.standard_normal((3, y1 - y0, W)).astype(np.float32)
            gw_hip[k] = gw[k]
        f = c_oracle.Forward(means3D=raw_cpu["xyz"], opacities=opac_a,
                             viewmatrix=cam.world_view_transform.contiguous().cpu().numpy(),
                             projmatrix=cam.full_proj_transform.cpu().numpy(),
                             campos=cam.camera_center.contiguous().cpu().numpy(), bg=bg.cpu().numpy(), width=W, height=H,
                             tanfovx=settings[k].tanfovx, tanfovy=settings[k].tanfovy, shs=shs_np, sh_degree=3,
                             scales=scales_a, rotations=rots_a, tile_rows=(r0, r1))
        d = np.abs(hip["color"][k][:, y0:y1].cpu().numpy() - f.out_color[:, y0:y1])
        view = {"image_max_abs": float(d.max()), "pixels_off_by_1e-4": float((d > 1e-4).mean()),
                "radii_mismatch": int((hip["radii"][k].cpu().numpy() != f.radii).sum()),
                "visible": int((f.radii > 0).sum())}
        if st is not None:
            view["final_T_equal"] = bool(np.array_equal(st["final_T"][k][y0:y1].cpu().numpy(), f.final_T[y0:y1]))
        res["views"].append(view)
        gb = f.backward(gw[k])
        for name in acc:
            acc[name] = acc[name] + np.asarray(gb[name], np.float64)
        f.close()
    # oracle gradients w.r.t. the RAW parameters: chain rule through exp / normalize / sigmoid
    gs, gq, go = c_oracle.activate_backward(raw_cpu["rotation"], scales_a, rots_a, opac_a, acc["scales"], acc["rotations"],
                                            acc["opacities"])
    ref = {"xyz": acc["means3D"], "scaling": gs, "rotation": gq, "opacity": go,
           "features": np.ascontiguousarray(acc["shs"].reshape(P, 48).T)}              # (P,16,3) -> (48,P) planar
    e = lambda like: torch.full_like(like, float("nan"))
    out = dict(means3D=e(v["xyz"]), sh=e(v["features"]), opacities=e(v["opacity"]), scales=e(v["scaling"]), rots=e(v["rotation"]))
    rasterizer.backward_multi(hip, torch.from_numpy(gw_hip).to(dev), out,
                              rank1={k: rasterizer.LUV_WEIGHTS for k in rank1_views} or None)
    got = {"xyz": out["means3D"], "scaling": out["scales"], "rotation": out["rots"], "opacity": out["opacities"],
           "features": out["sh"]}
    res["grad"] = {}
    for name, t in got.items():
        a = t.cpu().numpy().astype(np.float64)
        assert np.isfinite(a).all(), name
        b = np.asarray(ref[name], np.float64).reshape(a.shape)
        if name == "features":
            a, b = a.T, b.T                                                              # rows = Gaussians
        a2, b2 = a.reshape(P, -1), b.reshape(P, -1)
        scale = 1e-3 * np.abs(b2).max()
        per = np.abs(a2 - b2).max(axis=1) / (np.abs(b2).max(axis=1) + scale)
        res["grad"][name] = {"rel_l2": rel_l2(a2, b2), "per_gaussian_max": float(per.max()),
                             "per_gaussian_q999": float(np.quantile(per, 0.999)),
                             "per_gaussian_over_1e-3": int((per > 1e-3).sum())}
    return res


def assert_trainer_path_parity(res, grad_l2=1e-3, grad_pg=1e-3):
    """The operator's contract on the fused (PREACT, multi-view) path: no exceptions."""
    for v in res["views"]:
        assert v["radii_mismatch"] == 0, v
        assert v["image_max_abs"] == 0.0, v
        assert v.get("final_T_equal", True), v
    for name, g in res["grad"].items():
        assert g["rel_l2"] <= grad_l2, (name, g)
        assert g["per_gaussian_max"] <= grad_pg and g["per_gaussian_over_1e-3"] == 0, (name, g)


def assert_window_parity(res, grad_l2=1e-3, grad_pg=1e-3):
    assert res["radii_equal"], "radii differ from the oracle"
    assert res["image_max_abs"] == 0.0, res["image_max_abs"]            # bit-exact forward (DESIGN: arithmetic contract)
    assert res["final_T_equal"]
    assert res["hip_instances"] <= res["oracle_instances"]                 # exact tile culling only ever drops instances
    if res["hip_instances"] == res["oracle_instances"]:                    # n_contrib counts list POSITIONS: equal lists only
        assert res["n_contrib_equal"]
    assert res["n_contrib_mapped_equal"]        # culled lists: positions mapped back to the oracle's lists
    for name, (l2, pg) in res["grad"].items():
        assert l2 <= grad_l2, (name, "rel L2", l2)
        assert pg <= grad_pg, (name, "per-Gaussian", pg)
