"""Pure-PyTorch differentiable restatement of the Gaussian rasteriser (CPU, autograd).

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).

Purpose: an implementation that is independent of oracle/gs_oracle.c -- vectorised
per tile, gradients from autograd instead of hand-derived formulas, runs in fp32 or
fp64 -- so that the C oracle's analytic backward can be checked
against autograd + finite differences, and so that BASELINE config #1 ("5k Gaussians,
256x256, pure-PyTorch CPU path") has something to run.

PARITY UNPINNED for the rasteriser core (same reason as gs_oracle.c: the CUDA
submodule is absent from /root/reference).  Follows SURVEY Appendix A (forward) and
the three gradient conventions of Appendix B that differ from naive autograd:
  * min(0.99, alpha) is straight-through;
  * the guard-band clamp of t.x/t.z masks d/dt.x and treats the clamped t.x as a
    constant in d/dt.z;
  * (the 1/(det^2+1e-7) regulariser is NOT reproduced; <= 1.3e-5 relative.)
Reference anchors: gaussian_renderer/__init__.py:38-51,89-97,238-241;
scene/cameras.py:54-57; scene/gaussian_model.py:27-31; utils/general_utils.py:78-110;
utils/sh_utils.py:57-112.
"""
import math

import numpy as np
import torch

TILE = 16
NEAR_CULL_Z = 0.2
GUARD_BAND = 1.3
DILATION = 0.3
EIGEN_FLOOR = 0.1
ALPHA_CLAMP = 0.99
ALPHA_SKIP = 1.0 / 255.0
T_STOP = 1e-4
W_EPS = 1e-7

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]
SH_C4 = [2.5033429417967046, -1.7701307697799304, 0.9461746957575601, -0.6690465435572892, 0.10578554691520431,
         -0.6690465435572892, 0.47308734787878004, -1.7701307697799304, 0.6258357354491761]


def _now():
    import time
    return time.perf_counter()


def eval_sh_colors(deg, shs, dirs):
    """shs (P,M,3), dirs (P,3) unit -> (P,3) = max(sum_k Y_k sh_k + 0.5, 0).  utils/sh_utils.py:57-112."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    r = SH_C0 * shs[:, 0]
    if deg > 0:
        r = r - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = (r + SH_C2[0] * xy * shs[:, 4] + SH_C2[1] * yz * shs[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * shs[:, 6]
             + SH_C2[3] * xz * shs[:, 7] + SH_C2[4] * (xx - yy) * shs[:, 8])
    if deg > 2:
        r = (r + SH_C3[0] * y * (3 * xx - yy) * shs[:, 9] + SH_C3[1] * xy * z * shs[:, 10]
             + SH_C3[2] * y * (4 * zz - xx - yy) * shs[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12]
             + SH_C3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + SH_C3[5] * z * (xx - yy) * shs[:, 14]
             + SH_C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    if deg > 3:                                              # utils/sh_utils.py:97-110
        r = (r + SH_C4[0] * xy * (xx - yy) * shs[:, 16] + SH_C4[1] * yz * (3 * xx - yy) * shs[:, 17]
             + SH_C4[2] * xy * (7 * zz - 1) * shs[:, 18] + SH_C4[3] * yz * (7 * zz - 3) * shs[:, 19]
             + SH_C4[4] * (zz * (35 * zz - 30) + 3) * shs[:, 20] + SH_C4[5] * xz * (7 * zz - 3) * shs[:, 21]
             + SH_C4[6] * (xx - yy) * (7 * zz - 1) * shs[:, 22] + SH_C4[7] * xz * (xx - 3 * yy) * shs[:, 23]
             + SH_C4[8] * (xx * (xx - 3 * yy) - yy * (3 * xx - yy)) * shs[:, 24])
    return torch.clamp_min(r + 0.5, 0.0)


def build_cov3d(scales, rotations, mod):
    """(P,3),(P,4)->(P,6) [xx,xy,xz,yy,yz,zz]; quaternion taken as is (no normalisation)."""
    r, x, y, z = rotations.unbind(-1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    L = R * (mod * scales)[:, None, :]
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1)


def rasterize(means3D, opacities, *, viewmatrix, projmatrix, campos, bg, width, height, tanfovx, tanfovy,
              colors_precomp=None, shs=None, sh_degree=0, scales=None, rotations=None, cov3D_precomp=None,
              scale_modifier=1.0, means2D=None, return_aux=False, tile_rows=None, timings=None):
    """Differentiable forward.  Returns (color (3,H,W), radii (P,) int32[, aux dict]).

    If `means2D` (P,3, requires_grad) is given it receives the NDC-unit screen-space
    gradient exactly like the reference's `screenspace_points`
    (gaussian_renderer/__init__.py:28-32).
    `tile_rows=(r0, r1)`: only tile rows [r0, r1) are binned and composited (pixels outside keep 0);
    `timings`: dict that receives perf_counter stamps of the stages.
    """
    dt = means3D.dtype
    P = means3D.shape[0]
    W, H = int(width), int(height)
    if timings is not None:
        timings["start"] = _now()
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    V = viewmatrix.to(dt).reshape(4, 4)   # row-vector layout: p_row @ V
    Pm = projmatrix.to(dt).reshape(4, 4)
    bg = bg.to(dt)
    ones = torch.ones(P, 1, dtype=dt)
    hom = torch.cat([means3D, ones], 1)
    p_view = hom @ V[:, :3]
    p_hom = hom @ Pm
    p_w = 1.0 / (p_hom[:, 3] + W_EPS)
    ndc = p_hom[:, :2] * p_w[:, None]
    if means2D is not None:
        ndc = ndc + means2D[:, :2]   # value 0; carries d/d(ndc)
    vz = p_view[:, 2]
    in_front = vz.detach() > NEAR_CULL_Z

    S = cov3D_precomp.to(dt) if cov3D_precomp is not None else build_cov3d(scales, rotations, scale_modifier)
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    limx, limy = GUARD_BAND * tanfovx, GUARD_BAND * tanfovy
    tz = torch.where(in_front, vz, torch.ones_like(vz))
    def guard(t, lim):
        ratio = (t / tz).detach()
        inb = (ratio >= -lim) & (ratio <= lim)
        tc = (torch.clamp(ratio, -lim, lim) * tz.detach())
        return torch.where(inb, t + (tc - t).detach(), tc)
    tx, ty = guard(p_view[:, 0], limx), guard(p_view[:, 1], limy)
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], -1).reshape(-1, 2, 3)
    Wr = V[:3, :3].t()            # Wr[r][k] = V[k][r]
    T = J @ Wr
    S3 = torch.stack([S[:, 0], S[:, 1], S[:, 2], S[:, 1], S[:, 3], S[:, 4], S[:, 2], S[:, 4], S[:, 5]], -1).reshape(-1, 3, 3)
    cov2 = T @ S3 @ T.transpose(1, 2)
    a, b, c = cov2[:, 0, 0] + DILATION, cov2[:, 0, 1], cov2[:, 1, 1] + DILATION
    det = a * c - b * b
    det_ok = det.detach() != 0
    det_s = torch.where(det_ok, det, torch.ones_like(det))
    conic = torch.stack([c / det_s, -b / det_s, a / det_s], -1)
    mid = 0.5 * (a + c)
    disc = torch.sqrt(torch.clamp_min(mid * mid - det, EIGEN_FLOOR))
    lam = torch.maximum(mid + disc, mid - disc).detach()
    radius = torch.ceil(3.0 * torch.sqrt(lam)).to(torch.int64)
    pix = torch.stack([((ndc[:, 0].double() + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1].double() + 1.0) * H - 1.0) * 0.5], -1).to(dt)
    pd, rf = pix.detach(), radius.to(dt)
    trunc = lambda v: torch.trunc(v).to(torch.int64)
    xmin = trunc((pd[:, 0] - rf) / TILE).clamp(0, gx); ymin = trunc((pd[:, 1] - rf) / TILE).clamp(0, gy)
    xmax = trunc((pd[:, 0] + rf + TILE - 1) / TILE).clamp(0, gx); ymax = trunc((pd[:, 1] + rf + TILE - 1) / TILE).clamp(0, gy)
    area = (xmax - xmin) * (ymax - ymin)
    visible = in_front & det_ok & (area > 0)
    radii = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)

    if shs is not None:
        d = means3D - campos.to(dt)[None]
        d = d / d.norm(dim=1, keepdim=True)
        colors = eval_sh_colors(sh_degree, shs.to(dt), d)
    else:
        colors = colors_precomp.to(dt)
    opac = opacities.reshape(-1).to(dt)

    if timings is not None:
        timings["preprocess_done"] = _now()
    # ---- binning: instances sorted by (tile, depth bits, index) ----
    # vectorised: instance k of Gaussian g covers tile (ymin + k // w, xmin + k % w); with a tile-row window only the
    # rows [r0, r1) of each rectangle are generated (bench.py's bounded CPU sample)
    r0, r1 = (0, gy) if tile_rows is None else (max(0, int(tile_rows[0])), min(gy, int(tile_rows[1])))
    vis = torch.nonzero(visible).reshape(-1).numpy()
    xmin_n, xmax_n = xmin.numpy()[vis], xmax.numpy()[vis]
    ylo_n, yhi_n = np.maximum(ymin.numpy()[vis], r0), np.minimum(ymax.numpy()[vis], r1)
    wid = xmax_n - xmin_n
    cnt = wid * np.maximum(yhi_n - ylo_n, 0)
    depth_n = vz.detach().to(torch.float32).numpy()
    if cnt.sum() > 0:
        rep = np.repeat(np.arange(vis.shape[0]), cnt)
        k = np.arange(rep.shape[0]) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        tiles = (ylo_n[rep] + k // wid[rep]) * gx + xmin_n[rep] + k % wid[rep]
        gids = vis[rep].astype(np.int64)
    else:
        tiles, gids = np.zeros(0, np.int64), np.zeros(0, np.int64)
    order = np.lexsort((gids, depth_n[gids].view(np.uint32), tiles))
    tiles, gids = tiles[order], gids[order]
    starts = np.searchsorted(tiles, np.arange(gx * gy), "left")
    ends = np.searchsorted(tiles, np.arange(gx * gy), "right")

    out = torch.zeros(3, H, W, dtype=dt)
    final_T = torch.ones(H, W, dtype=dt)
    n_contrib = torch.zeros(H, W, dtype=torch.int64)
    rows = []
    if timings is not None:
        timings["binning_done"] = _now()
    for t in range(r0 * gx, r1 * gx):
        tyi, txi = divmod(t, gx)
        y0, x0 = tyi * TILE, txi * TILE
        y1, x1 = min(y0 + TILE, H), min(x0 + TILE, W)
        ys, xs = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
        npx = ys.numel()
        if ends[t] == starts[t]:
            out[:, y0:y1, x0:x1] = bg[:, None, None]
            continue
        ids = torch.from_numpy(gids[starts[t]:ends[t]])
        pf = torch.stack([xs.reshape(-1), ys.reshape(-1)], -1).to(dt)          # (npx,2)
        dxy = pix[ids][:, None, :] - pf[None]                                  # (n,npx,2)
        con = conic[ids]
        power = -0.5 * (con[:, 0, None] * dxy[..., 0] ** 2 + con[:, 2, None] * dxy[..., 1] ** 2) - con[:, 1, None] * dxy[..., 0] * dxy[..., 1]
        G = torch.exp(torch.clamp_max(power, 0.0))
        araw = opac[ids][:, None] * G
        alpha = araw + (torch.clamp_max(araw, ALPHA_CLAMP) - araw).detach()
        valid = (power.detach() <= 0) & (alpha.detach() >= ALPHA_SKIP)
        a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
        om = 1.0 - a_eff
        T_incl = torch.cumprod(om, 0)
        T_excl = torch.cat([torch.ones(1, npx, dtype=dt), T_incl[:-1]], 0)
        dropped = torch.cumsum((valid & (T_incl.detach() < T_STOP)).to(torch.int64), 0) > 0
        keep = valid & ~dropped
        w = torch.where(keep, a_eff * T_excl, torch.zeros_like(a_eff))
        C = torch.einsum("np,nc->cp", w, colors[ids])
        Tf = torch.where(keep, om, torch.ones_like(om)).prod(0)
        out[:, y0:y1, x0:x1] = (C + Tf[None] * bg[:, None]).reshape(3, y1 - y0, x1 - x0)
        final_T[y0:y1, x0:x1] = Tf.detach().reshape(y1 - y0, x1 - x0)
        idx1 = torch.arange(1, ids.numel() + 1)[:, None] * keep
        n_contrib[y0:y1, x0:x1] = idx1.max(0).values.reshape(y1 - y0, x1 - x0)
    if timings is not None:
        timings["composite_done"] = _now()
    if not return_aux:
        return out, radii
    aux = {"xy": pix.detach(), "conic": conic.detach(), "depth": vz.detach(), "rect": torch.stack([xmin, ymin, xmax, ymax], -1),
           "tiles_touched": torch.where(visible, area, torch.zeros_like(area)), "point_list": gids, "tile_list": tiles,
           "ranges": np.stack([starts, ends], -1), "final_T": final_T, "n_contrib": n_contrib, "cov3d": S.detach(),
           "colors": colors.detach(), "num_rendered": int(tiles.shape[0]),
           # the per-Gaussian tensors the compositing reads, still attached to the graph: lets a caller time / inspect the
           # compositing backward (d/d these) apart from the per-Gaussian backward (these -> inputs)
           "diff": (pix, conic, colors, opac)}
    return out, radii, aux


# ---------------------------------------------------------------------------
# Python-side pieces of the path, restated for the CPU.
# ---------------------------------------------------------------------------
def luminance(img):
    """utils/loss_utils.py:24-28"""
    return (0.4124 * img[0] + 0.35758 * img[1] + 0.1804 * img[2]).unsqueeze(0)


def event_frame(img_now, img_next, C):
    """utils/loss_utils.py:234-249"""
    return (torch.log(luminance(img_next) + 1e-8) - torch.log(luminance(img_now) + 1e-8)) / C


def event_iteration_loss(image, img_now, img_next, gt_int, gt_now, gt_next, c, gt_blur=None, gt_c=0.17):
    """train.py:165-203 (lambda_dssim forced to 0 at :177; the 0*ssim term is dropped)."""
    img_diff = event_frame(img_now, img_next, c)
    gt = event_frame(gt_now, gt_next, gt_c)
    loss1 = torch.abs(img_diff - gt).mean()
    loss2 = torch.abs(image - gt_int).mean()
    mask = (gt != 0).to(image.dtype)
    loss = 0.9 * (loss1 * mask).sum() + 0.1 * (loss2 * (1 - mask)).sum()
    loss = loss / (mask.sum() + (1 - mask).sum())
    if gt_blur is not None:
        loss = 0.5 * loss + 0.5 * torch.abs(image - gt_blur).mean()
    return loss


def _ssim_window(channel, dtype):
    """utils/loss_utils.py:359-367"""
    g = torch.tensor([math.exp(-(x - 5) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)], dtype=torch.float32)
    g = (g / g.sum()).unsqueeze(1)
    w2 = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, 11, 11).contiguous().to(dtype)


def ssim(img1, img2):
    """utils/loss_utils.py:388-418"""
    import torch.nn.functional as F
    C = img1.shape[-3]
    w = _ssim_window(C, img1.dtype)
    a, b = img1.unsqueeze(0), img2.unsqueeze(0)
    mu1, mu2 = F.conv2d(a, w, padding=5, groups=C), F.conv2d(b, w, padding=5, groups=C)
    s11 = F.conv2d(a * a, w, padding=5, groups=C) - mu1 * mu1
    s22 = F.conv2d(b * b, w, padding=5, groups=C) - mu2 * mu2
    s12 = F.conv2d(a * b, w, padding=5, groups=C) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))
    return m.mean()


def to_gray(img):
    """utils/loss_utils.py:18-23"""
    return (0.299 * img[0] + 0.587 * img[1] + 0.114 * img[2]).unsqueeze(0)


def gray_iteration_loss(image, gt, lambda_dssim=0.2):
    """train.py:213-223"""
    l1 = torch.abs(to_gray(image) - to_gray(gt)).mean()
    return (1.0 - lambda_dssim) * l1 + lambda_dssim * (1.0 - ssim(to_gray(image), to_gray(gt)))


def look_at_camera(eye, target, up, fovx, width, height, znear=0.01, zfar=100.0, dtype=torch.float32):
    """Builds (viewmatrix, projmatrix, campos, tanfovx, tanfovy) in the reference's
    row-vector layout: world_view_transform = W2C^T, full_proj = W2C^T @ P^T
    (scene/cameras.py:54-57; utils/graphics_utils.py:38-71)."""
    eye, target, up = (np.asarray(v, np.float64) for v in (eye, target, up))
    fwd = target - eye; fwd /= np.linalg.norm(fwd)            # camera +z looks at the scene
    right = np.cross(fwd, up); right /= np.linalg.norm(right)  # camera +x
    down = np.cross(fwd, right)                                # camera +y (image y grows downward)
    Rw2c = np.stack([right, down, fwd], 0)
    W2C = np.eye(4); W2C[:3, :3] = Rw2c; W2C[:3, 3] = -Rw2c @ eye
    W2C = np.float32(W2C)
    fovy = 2.0 * math.atan(math.tan(fovx / 2.0) * height / width)
    tx, ty = math.tan(fovx / 2.0), math.tan(fovy / 2.0)
    Pj = torch.zeros(4, 4)
    top, right_ = ty * znear, tx * znear
    Pj[0, 0] = 2.0 * znear / (2 * right_); Pj[1, 1] = 2.0 * znear / (2 * top)
    Pj[3, 2] = 1.0; Pj[2, 2] = zfar / (zfar - znear); Pj[2, 3] = -(zfar * znear) / (zfar - znear)
    view = torch.tensor(W2C).transpose(0, 1).contiguous()
    proj = (view.unsqueeze(0).bmm(Pj.transpose(0, 1).unsqueeze(0))).squeeze(0)
    campos = view.inverse()[3, :3].contiguous()
    return view.to(dtype), proj.to(dtype), campos.to(dtype), tx, ty


def psnr(img1, img2):
    """utils/image_utils.py:19-21"""
    mse = ((img1 - img2) ** 2).reshape(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


def eval_gray_psnr(render_fn, cameras_with_gt, index_list=(5, 25, 45, 65, 85)):
    """eval.py:118-152 (the PSNR half): clamp, rgb_to_grayscale on render and ground truth, mean over the held-out views.
    `cameras_with_gt`: list of (camera, gt (3,H,W) CPU tensor)."""
    total = 0.0
    for index in index_list:
        cam, gt = cameras_with_gt[index]
        img = to_gray(torch.clamp(render_fn(cam), 0.0, 1.0))
        total += float(psnr(img, to_gray(torch.clamp(gt, 0.0, 1.0))).mean())
    return total / len(index_list)
