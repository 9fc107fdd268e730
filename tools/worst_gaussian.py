"""Diagnostic (GPU box): where does the worst per-Gaussian gradient difference of the timed path come from?
Runs helpers.trainer_path_parity's comparison on the cfg3 scene, one view at a time with an explicit grad_acc, and
prints, for the worst Gaussian of each gradient, the per-splat sums (HIP fp32 vs oracle double-accumulated) next to the
final gradients.   python tools/worst_gaussian.py [N]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import torch
from event_3dgs_amd import rasterizer, _lib
from event_3dgs_amd.train_step import EventTrainer
from oracle import c_oracle
from test_hip_configs import _setup

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
W, H = 1920, 1080
params, cams, bg, gts = _setup(N, W, H)
tr = EventTrainer(params, "cuda:0")
rows = [(32, 34), (0, 2), (66, 68)]
raw_cpu = {k: t.detach().cpu().numpy() for k, t in tr.views.items()}
sc, ro, op = c_oracle.activate(raw_cpu["scaling"], raw_cpu["rotation"], raw_cpu["opacity"])
shs = np.ascontiguousarray(raw_cpu["features"].T.reshape(-1, 16, 3))
v = tr.views
rng = np.random.default_rng(11)
for k, (cam, (r0, r1)) in enumerate(zip(cams, rows)):
    st = [tr._settings(cam, bg)]
    hip = rasterizer.forward_multi(v["xyz"], v["features"], v["opacity"], v["scaling"], v["rotation"], st, flags=tr.FWD_FLAGS)
    y0, y1 = r0 * 16, min(H, r1 * 16)
    gw = np.zeros((1, 3, H, W), np.float32)
    gw[0, :, y0:y1] = rng.standard_normal((3, y1 - y0, W)).astype(np.float32)
    f = c_oracle.Forward(means3D=raw_cpu["xyz"], opacities=op, viewmatrix=cam.world_view_transform.contiguous().cpu().numpy(),
                         projmatrix=cam.full_proj_transform.cpu().numpy(), campos=cam.camera_center.contiguous().cpu().numpy(),
                         bg=bg.cpu().numpy(), width=W, height=H, tanfovx=st[0].tanfovx, tanfovy=st[0].tanfovy, shs=shs,
                         sh_degree=3, scales=sc, rotations=ro, tile_rows=(r0, r1))
    gb = f.backward(gw[0])
    I = hip["num_rendered"]
    acc = torch.zeros(I + N, _lib.ACC_STRIDE, device="cuda:0")
    e = lambda like: torch.full_like(like, float("nan"))
    out = dict(means3D=e(v["xyz"]), sh=e(v["features"]), opacities=e(v["opacity"]), scales=e(v["scaling"]), rots=e(v["rotation"]),
               means2D=torch.zeros(N, 3, device="cuda:0"))
    rasterizer.backward_multi(hip, torch.from_numpy(gw).cuda(), out, grad_acc=acc)
    sums = acc[I:I + N].cpu().numpy().astype(np.float64)       # mx my A B | C o c0 c1 | c2
    gs, gq, go = c_oracle.activate_backward(raw_cpu["rotation"], sc, ro, op, gb["scales"], gb["rotations"], gb["opacities"])
    a = out["scales"].cpu().numpy().astype(np.float64)
    per = np.abs(a - gs).max(axis=1) / (np.abs(gs).max(axis=1) + 1e-3 * np.abs(gs).max())
    vis = f.radii > 0
    conic_h = np.stack([sums[:, 2], sums[:, 3], sums[:, 4]], 1)
    conic_o = gb["conic"].astype(np.float64)
    cs = 1e-3 * np.abs(conic_o).max()
    per_c = np.abs(conic_h - conic_o).max(axis=1) / (np.abs(conic_o).max(axis=1) + cs)
    per_c[~vis] = 0
    print(f"view {k}: scaling per-Gaussian max {per.max():.3e} (#>1e-3: {(per > 1e-3).sum()}, #>3e-4: {(per > 3e-4).sum()}); "
          f"conic sums per-Gaussian max {per_c.max():.3e}")
    for i in np.argsort(-per)[:3]:
        print(f"  g={i} per={per[i]:.3e} radius={f.radii[i]} scale={sc[i]} opac={op[i, 0]:.4f}")
        print(f"     scales HIP {a[i]}  oracle {gs[i]}  max|ref| {np.abs(gs).max():.3e}")
        print(f"     conic  HIP {conic_h[i]}  oracle {conic_o[i]}  rel {np.abs(conic_h[i] - conic_o[i]) / (np.abs(conic_o[i]) + 1e-30)}")
        print(f"     mean2D HIP {sums[i, :2]}  oracle {gb['means2D'][i, :2]}   conic_opacity {f.conic_opacity[i]}")
    f.close()
