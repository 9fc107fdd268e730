"""Generates tests/golden/*.npz by IMPORTING the reference's own Python (this container only).

Run:  python tests/golden/make_golden.py          (needs /root/reference; never runs on the GPU box)

Only arrays (inputs + the reference's outputs) are written; no reference source
travels.  Recipe = SURVEY.md Appendix E: stub the modules the imported functions
never touch, and map device "cuda" to CPU.

Vectors:
  G1 sh.npz          utils/sh_utils.py eval_sh deg 0..3 + clamp of gaussian_renderer/__init__.py:81, with autograd grads
  G2 cov3d.npz       scene/gaussian_model.py:27-31 build_covariance_from_scaling_rotation + grads
  G3 cameras.npz     scene/cameras.py Camera matrices for 4 seeded poses (one 1920x1080)
  G4 event_loss.npz  utils/loss_utils.py differentialable_event_simu / l1_loss + the train.py:165-203 composition, grads
  G5 image_metrics.npz  ssim, ssim_gray, l1_loss_gray, psnr values
  G7 lr.npz          utils/general_utils.py get_expon_lr_func at steps {0,1,100,7000,30000}
  G8 densify.npz     scene/gaussian_model.py densify_and_prune + reset_opacity on a seeded 256-Gaussian model (torch.manual_seed(77))
  G10 projection.npz  gaussian_renderer/__init__.py project_points / generate_depth_map (the reference's CPU restatement of
                     the rasteriser's projection + ndc2Pix) on seeded points under two of the G3 cameras
  G9 colmap_tiny/ + colmap_tiny.npz   a tiny COLMAP model (bin + txt, written here) as parsed by scene/colmap_loader.py,
                     plus the derived R/T/FoV (scene/dataset_readers.py:84-97) and getNerfppNorm (:47-68)
"""
import os
import sys
import types

import numpy as np
import torch
from math import log as math_log

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    for n in ["torchvision", "torchgeometry", "lpips", "plyfile", "simple_knn", "simple_knn._C",
              "diff_gaussian_rasterization"]:
        sys.modules[n] = types.ModuleType(n)
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    sys.modules["simple_knn._C"].distCUDA2 = None
    sys.modules["diff_gaussian_rasterization"].GaussianRasterizationSettings = None
    sys.modules["diff_gaussian_rasterization"].GaussianRasterizer = None
    torch.Tensor.cuda = lambda self, *a, **k: self

    def _wrap(fn):
        def inner(*a, **k):
            if k.get("device") is not None and "cuda" in str(k["device"]):
                k["device"] = "cpu"
            return fn(*a, **k)
        return inner
    for name in ["zeros", "ones", "tensor", "empty", "rand", "zeros_like", "ones_like", "full"]:
        setattr(torch, name, _wrap(getattr(torch, name)))


def main():
    _import_reference()
    from utils.sh_utils import eval_sh, RGB2SH
    from utils.general_utils import build_scaling_rotation, strip_symmetric, get_expon_lr_func
    from utils.loss_utils import (differentialable_event_simu, l1_loss, l1_loss_gray, ssim, ssim_gray)
    from utils.image_utils import psnr
    from scene.cameras import Camera

    g = torch.Generator().manual_seed(1234)

    # ---- G1: SH colours as render() computes them (gaussian_renderer/__init__.py:74-81) ----
    out = {}
    N = 64
    feats = torch.randn(N, 16, 3, generator=g) * 0.4          # (N,16,3) like pc.get_features
    xyz = torch.randn(N, 3, generator=g)
    campos = torch.tensor([0.3, -0.2, 2.5])
    out["features"], out["xyz"], out["campos"] = feats.numpy(), xyz.numpy(), campos.numpy()
    gcol = torch.randn(N, 3, generator=g)
    out["grad_colors"] = gcol.numpy()
    for deg in range(4):
        f = feats.clone().requires_grad_(True)
        p = xyz.clone().requires_grad_(True)
        shs_view = f.transpose(1, 2).view(-1, 3, 16)
        dir_pp = p - campos.repeat(N, 1)
        dirn = dir_pp / dir_pp.norm(dim=1, keepdim=True)
        col = torch.clamp_min(eval_sh(deg, shs_view, dirn) + 0.5, 0.0)
        (col * gcol).sum().backward()
        out[f"colors_deg{deg}"] = col.detach().numpy()
        out[f"dfeatures_deg{deg}"] = f.grad.numpy()
        out[f"dxyz_deg{deg}"] = p.grad.numpy() if p.grad is not None else np.zeros((N, 3), np.float32)
    out["rgb2sh_of_half"] = RGB2SH(torch.tensor([0.5, 0.0, 1.0])).numpy()
    np.savez(os.path.join(OUT, "sh.npz"), **out)

    # ---- G2: covariance twin (scene/gaussian_model.py:27-31) ----
    s = torch.exp(torch.randn(N, 3, generator=g) * 0.7 - 2).requires_grad_(True)
    q = torch.randn(N, 4, generator=g)
    q = (q / q.norm(dim=1, keepdim=True)).requires_grad_(True)
    gc = torch.randn(N, 6, generator=g)
    for mod in (1.0, 0.7):
        s.grad = q.grad = None
        L = build_scaling_rotation(mod * s, q)
        cov = strip_symmetric(L @ L.transpose(1, 2))
        (cov * gc).sum().backward()
        tag = "" if mod == 1.0 else "_mod07"
        np.savez(os.path.join(OUT, f"cov3d{tag}.npz"), scales=s.detach().numpy(), rotations=q.detach().numpy(),
                 cov=cov.detach().numpy(), grad_cov=gc.numpy(), dscales=s.grad.numpy(), drotations=q.grad.numpy(),
                 mod=np.float32(mod))

    # ---- G3: cameras ----
    cams = {}
    rs = np.random.RandomState(7)
    sizes = [(48, 32), (800, 800), (1920, 1080), (333, 217)]
    for k, (w, h) in enumerate(sizes):
        A = rs.randn(3, 3)
        Q, _ = np.linalg.qr(A)
        if np.linalg.det(Q) < 0:
            Q[:, 0] *= -1
        T = rs.randn(3) * 2
        fovx = 0.6911112070083618 if k != 3 else 1.1
        fovy = 2 * np.arctan(np.tan(fovx / 2) * h / w)
        cam = Camera(colmap_id=k, R=Q, T=T, FoVx=fovx, FoVy=fovy, image=torch.zeros(3, h, w), gt_alpha_mask=None,
                     image_name=str(k), uid=k, data_device="cpu")
        cams[f"R{k}"], cams[f"T{k}"] = Q, T
        cams[f"fov{k}"] = np.array([fovx, fovy]); cams[f"size{k}"] = np.array([w, h])
        cams[f"view{k}"] = cam.world_view_transform.contiguous().numpy()
        cams[f"viewstride{k}"] = np.array(cam.world_view_transform.stride())
        cams[f"proj{k}"] = cam.full_proj_transform.contiguous().numpy()
        cams[f"center{k}"] = cam.camera_center.contiguous().numpy()
        cams[f"P{k}"] = cam.projection_matrix.contiguous().numpy()
    np.savez(os.path.join(OUT, "cameras.npz"), **cams)

    # ---- G10: the reference's own CPU projection (gaussian_renderer/__init__.py:194-273: project_points and the
    # ndc2Pix line + int() truncation of generate_depth_map) on seeded points in front of two of the cameras above.
    # It is the only restatement of the rasteriser's projection the reference tree holds (eps 1e-4 instead of 1e-7).
    import gaussian_renderer as GR
    pj = {}
    g10 = torch.Generator().manual_seed(4321)        # own stream: the vectors below keep their seeds
    for k in (0, 3):
        w, h = sizes[k]
        view = torch.tensor(cams[f"view{k}"]); proj = torch.tensor(cams[f"proj{k}"]); centre = torch.tensor(cams[f"center{k}"])
        fovx, fovy = cams[f"fov{k}"]
        z = torch.rand(300, 1, generator=g10) * 5.0 + 1.0
        xy = (torch.rand(300, 2, generator=g10) * 2 - 1) * torch.tensor([np.tan(fovx / 2), np.tan(fovy / 2)]).float() * 1.1 * z
        pv = torch.cat((xy, z, torch.ones(300, 1)), 1).double()
        pw = (pv @ torch.linalg.inv(view.double()))[:, :3].float().contiguous()
        ndc = GR.project_points(pw.numpy(), proj.numpy())
        depth = GR.generate_depth_map(pw, centre, proj, (w, h), None)
        pj[f"points{k}"], pj[f"ndc{k}"], pj[f"depth{k}"], pj[f"cam{k}"] = pw.numpy(), ndc, depth.numpy(), np.int64(k)
    np.savez(os.path.join(OUT, "projection.npz"), **pj)

    # ---- G4: event loss pieces + the train.py:165-203 composition ----
    H, W = 32, 48
    ev = {}
    img = (torch.rand(3, H, W, generator=g))
    now = torch.rand(3, H, W, generator=g)
    nxt = (now + 0.1 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
    now[:, :4, :] = 0.0                                        # black background region: Y = 0 -> ln(1e-8)
    nxt[:, :2, :] = 0.0
    gt_int = torch.rand(3, H, W, generator=g)
    q8 = lambda t: torch.round(t * 255) / 255                  # 8-bit GT like PIL images
    gt_now = q8(torch.rand(3, H, W, generator=g))
    gt_next = gt_now.clone()
    chg = torch.rand(H, W, generator=g) < 0.35                 # 35 % of pixels change -> rho ~ 0.35
    gt_next[:, chg] = q8(torch.rand(3, int(chg.sum()), generator=g))
    gt_blur = torch.rand(3, H, W, generator=g)
    for name, t in dict(image=img, now=now, next=nxt, gt_int=gt_int, gt_now=gt_now, gt_next=gt_next, gt_blur=gt_blur).items():
        ev[name] = t.numpy()
    for deblur in (False, True):
        c = torch.nn.Parameter(torch.tensor(0.17 if not deblur else 0.23))
        a, b_, d = img.clone().requires_grad_(True), now.clone().requires_grad_(True), nxt.clone().requires_grad_(True)
        img_diff = differentialable_event_simu(b_, d, False, c)
        gt_image = differentialable_event_simu(gt_now, gt_next, False, 0.17)
        Ll1 = l1_loss(img_diff, gt_image)
        lambda_dssim = 0
        loss1 = (1.0 - lambda_dssim) * Ll1 + lambda_dssim * (1.0 - ssim_gray(img_diff, gt_image))
        Ll1b = l1_loss(a, gt_int)
        loss2 = (1.0 - lambda_dssim) * Ll1b
        mask = (gt_image != 0).float()
        loss = 0.9 * (loss1 * mask).sum() + (1 - 0.9) * (loss2 * (1 - mask)).sum()
        loss /= (mask.sum() + (1 - mask).sum())
        if deblur:
            loss = (1.0 - 0.5) * loss + 0.5 * l1_loss(a, gt_blur)
        loss.backward()
        t = "_deblur" if deblur else ""
        ev["c" + t] = c.detach().numpy(); ev["loss" + t] = loss.detach().numpy()
        ev["d_image" + t], ev["d_now" + t], ev["d_next" + t] = a.grad.numpy(), b_.grad.numpy(), d.grad.numpy()
        ev["d_c" + t] = c.grad.numpy()
        ev["img_diff" + t] = img_diff.detach().numpy(); ev["gt_diff"] = gt_image.numpy()
        ev["rho"] = mask.mean().numpy()
    np.savez(os.path.join(OUT, "event_loss.npz"), **ev)

    # ---- G5: image metrics ----
    a = torch.rand(3, 40, 56, generator=g); b2 = (a + 0.1 * torch.randn(3, 40, 56, generator=g)).clamp(0, 1)
    a.requires_grad_(True)
    sv = ssim(a, b2); sg = ssim_gray(a, b2); lg = l1_loss_gray(a, b2)
    loss_gray = 0.8 * lg + 0.2 * (1.0 - sg)                    # train.py:213-223 with default lambda_dssim
    loss_gray.backward()
    np.savez(os.path.join(OUT, "image_metrics.npz"), a=a.detach().numpy(), b=b2.numpy(), ssim=sv.detach().numpy(),
             ssim_gray=sg.detach().numpy(), l1_gray=lg.detach().numpy(), psnr=psnr(a.detach(), b2).numpy(),
             gray_loss=loss_gray.detach().numpy(), d_a_gray_loss=a.grad.numpy())

    # ---- G7: LR schedule (arguments/__init__.py:77-81 defaults, scene extent 1) ----
    fn = get_expon_lr_func(lr_init=1.6e-4, lr_final=1.6e-6, lr_delay_mult=0.01, max_steps=30000)
    steps = np.array([0, 1, 100, 7000, 30000])
    np.savez(os.path.join(OUT, "lr.npz"), steps=steps, lr=np.array([fn(int(s)) for s in steps], np.float64))
    # ---- G8: one densify_and_prune + reset_opacity on a seeded 256-Gaussian model (scene/gaussian_model.py:258-403)
    from argparse import ArgumentParser
    from arguments import OptimizationParams
    from scene.gaussian_model import GaussianModel
    gm = GaussianModel(3)
    Nn = 256
    gg = torch.Generator().manual_seed(99)
    P_ = lambda t: torch.nn.Parameter(t.requires_grad_(True))
    gm._xyz = P_(torch.randn(Nn, 3, generator=gg))
    gm._features_dc = P_(torch.randn(Nn, 1, 3, generator=gg))
    gm._features_rest = P_(torch.randn(Nn, 15, 3, generator=gg) * 0.1)
    gm._scaling = P_(torch.randn(Nn, 3, generator=gg) * 0.8 + math_log(0.04))
    gm._rotation = P_(torch.randn(Nn, 4, generator=gg))
    gm._opacity = P_(torch.randn(Nn, 1, generator=gg) * 3.0)
    gm.max_radii2D = torch.zeros(Nn)
    gm.spatial_lr_scale = 1.0
    opt = OptimizationParams(ArgumentParser())
    gm.training_setup(opt)
    for grp in gm.optimizer.param_groups:                       # one Adam step so that the moments are non-trivial
        grp["params"][0].grad = torch.randn(grp["params"][0].shape, generator=gg) * 1e-3
    gm.optimizer.step()
    gm.xyz_gradient_accum = torch.rand(Nn, 1, generator=gg) * 6e-4
    gm.denom = torch.randint(0, 3, (Nn, 1), generator=gg).float()
    gm.max_radii2D = torch.rand(Nn, generator=gg) * 40
    names = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
    def dump(tag, d):
        for grp in gm.optimizer.param_groups:
            p_ = grp["params"][0]
            d[f"{tag}_{grp['name']}"] = p_.detach().numpy().copy()
            st = gm.optimizer.state[p_]
            d[f"{tag}_{grp['name']}_m"] = st["exp_avg"].numpy().copy()
            d[f"{tag}_{grp['name']}_v"] = st["exp_avg_sq"].numpy().copy()
        d[f"{tag}_accum"] = gm.xyz_gradient_accum.numpy().copy()
        d[f"{tag}_denom"] = gm.denom.numpy().copy()
        d[f"{tag}_maxr"] = gm.max_radii2D.numpy().copy()
    dd = {}
    dump("in", dd)
    torch.manual_seed(77)
    gm.densify_and_prune(0.0002, 0.005, 4.0, 20)
    dump("out", dd)
    gm.reset_opacity()
    dump("reset", dd)
    dd["args"] = np.array([0.0002, 0.005, 4.0, 20.0, 0.01])      # max_grad, min_opacity, extent, max_screen_size, percent_dense
    np.savez(os.path.join(OUT, "densify.npz"), **dd)
    # ---- G9: tiny COLMAP model (binary + text) written here with struct, parsed by the reference's colmap_loader
    import struct
    from scene import colmap_loader as CL
    from scene.dataset_readers import getNerfppNorm
    from utils.graphics_utils import focal2fov
    cdir = os.path.join(OUT, "colmap_tiny")
    os.makedirs(cdir, exist_ok=True)
    rs2 = np.random.RandomState(3)
    cams_spec = [(1, 1, 640, 480, [500.0, 510.0, 320.0, 240.0]), (2, 0, 800, 600, [700.0, 400.0, 300.0])]
    with open(os.path.join(cdir, "cameras.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(cams_spec)))
        for cid, mid, w, h, pr in cams_spec:
            f.write(struct.pack("<iiQQ", cid, mid, w, h)); f.write(struct.pack("<" + "d" * len(pr), *pr))
    imgs_spec = []
    for i in range(5):
        q = rs2.randn(4); q /= np.linalg.norm(q)
        imgs_spec.append((i + 1, q, rs2.randn(3) * 2, 1 + (i % 2), f"img_{4 - i:03d}.png", rs2.rand(3, 2) * 100, rs2.randint(-1, 50, 3)))
    with open(os.path.join(cdir, "images.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(imgs_spec)))
        for iid, q, t, cid, name, xys, pids in imgs_spec:
            f.write(struct.pack("<idddddddi", iid, *q, *t, cid)); f.write(name.encode() + b"\x00")
            f.write(struct.pack("<Q", len(pids)))
            for (x, y), pid in zip(xys, pids):
                f.write(struct.pack("<ddq", x, y, int(pid)))
    npts = 7
    pts = rs2.randn(npts, 3); cols = rs2.randint(0, 256, (npts, 3)); errs = rs2.rand(npts)
    with open(os.path.join(cdir, "points3D.bin"), "wb") as f:
        f.write(struct.pack("<Q", npts))
        for i in range(npts):
            f.write(struct.pack("<QdddBBBd", i + 1, *pts[i], *[int(c) for c in cols[i]], errs[i]))
            tl = i % 3
            f.write(struct.pack("<Q", tl)); f.write(struct.pack("<" + "ii" * tl, *([1, 2] * tl)))
    with open(os.path.join(cdir, "cameras.txt"), "w") as f:
        f.write("# Camera list\n")
        for cid, mid, w, h, pr in cams_spec:           # the reference's text reader accepts PINHOLE only (colmap_loader.py:171)
            pr4 = pr if mid == 1 else [pr[0], pr[0], pr[1], pr[2]]
            f.write(f"{cid} PINHOLE {w} {h} " + " ".join(repr(x) for x in pr4) + "\n")
    with open(os.path.join(cdir, "images.txt"), "w") as f:
        f.write("# Image list with two lines of data per image\n")
        for iid, q, t, cid, name, xys, pids in imgs_spec:
            f.write(f"{iid} " + " ".join(repr(float(x)) for x in list(q) + list(t)) + f" {cid} {name}\n")
            f.write(" ".join(f"{repr(float(x))} {repr(float(y))} {int(pid)}" for (x, y), pid in zip(xys, pids)) + "\n")
    with open(os.path.join(cdir, "points3D.txt"), "w") as f:
        f.write("# 3D point list\n")
        for i in range(npts):
            f.write(f"{i + 1} " + " ".join(repr(float(x)) for x in pts[i]) + " " + " ".join(str(int(c)) for c in cols[i]) + f" {repr(float(errs[i]))} 1 2\n")
    cb, ib = CL.read_intrinsics_binary(os.path.join(cdir, "cameras.bin")), CL.read_extrinsics_binary(os.path.join(cdir, "images.bin"))
    xb, rb, eb = CL.read_points3D_binary(os.path.join(cdir, "points3D.bin"))
    ct, it_ = CL.read_intrinsics_text(os.path.join(cdir, "cameras.txt")), CL.read_extrinsics_text(os.path.join(cdir, "images.txt"))
    xt, rt, et = CL.read_points3D_text(os.path.join(cdir, "points3D.txt"))
    cg = {"xyz_bin": xb, "rgb_bin": rb, "err_bin": eb, "xyz_txt": xt, "rgb_txt": rt, "err_txt": et}
    for tag, cc, ii in (("bin", cb, ib), ("txt", ct, it_)):
        for k, c in cc.items():
            cg[f"cam_{tag}_{k}"] = np.array([c.id, c.width, c.height] + list(c.params), np.float64); cg[f"cammodel_{tag}_{k}"] = np.array(c.model)
        for k, im in ii.items():
            cg[f"img_{tag}_{k}"] = np.concatenate([[im.id], im.qvec, im.tvec, [im.camera_id]]); cg[f"imgname_{tag}_{k}"] = np.array(im.name)
            cg[f"imgxys_{tag}_{k}"] = im.xys; cg[f"imgpids_{tag}_{k}"] = im.point3D_ids
    # derived view parameters (scene/dataset_readers.py:84-97) and the NeRF++ normalisation (:47-68)
    class CI:  # minimal CameraInfo
        pass
    infos = []
    for k, ex in ib.items():
        intr = cb[ex.camera_id]
        ci = CI(); ci.R = np.transpose(CL.qvec2rotmat(ex.qvec)); ci.T = np.array(ex.tvec)
        fx = intr.params[0]; fy = intr.params[1] if intr.model == "PINHOLE" else intr.params[0]
        cg[f"view_{k}"] = np.concatenate([ci.R.reshape(-1), ci.T, [focal2fov(fx, intr.width), focal2fov(fy, intr.height)]])
        infos.append(ci)
    nn_ = getNerfppNorm(infos)
    cg["nerfpp_translate"], cg["nerfpp_radius"] = nn_["translate"], np.float64(nn_["radius"])
    np.savez(os.path.join(OUT, "colmap_tiny.npz"), **cg)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
