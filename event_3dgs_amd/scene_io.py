"""Dataset directory loader for the reference's COLMAP layout and the evaluation
protocol of eval.py (8f-4).  Host code: PIL + numpy; images end up as (3,H,W) fp32 tensors on `device`.

    <scene>/images/          intensity frames            -> train cameras        (scene/dataset_readers.py:146-148)
    <scene>/images_event/    frames used for differences -> event cameras (--event, :154-159)
    <scene>/images_blurry/   blurry frames               -> blurry cameras (--deblur, :149-153)
    <scene>/renders/         ground truth                -> test cameras         (:174-179)
    <scene>/sparse/0/{cameras,images,points3D}.{bin,txt} (+ points3D.ply written on first load, :183-192)
"""
import collections
import os

import numpy as np
import torch

from . import io_formats as IO
from .cameras import Camera

SceneData = collections.namedtuple("SceneData", ["train_cameras", "event_cameras", "blurry_cameras", "test_cameras",
                                                 "point_cloud", "cameras_extent", "translate", "ply_path"])


def pil_to_torch(pil_image, resolution):
    """utils/general_utils.py:21-27"""
    resized = torch.from_numpy(np.array(pil_image.resize(resolution))) / 255.0
    return resized.permute(2, 0, 1) if resized.dim() == 3 else resized.unsqueeze(-1).permute(2, 0, 1)


def target_resolution(orig_w, orig_h, resolution=-1, resolution_scale=1.0):
    """utils/camera_utils.py:19-39: -r 1/2/4/8 are exact divisors (rounded); -1 caps the width at 1600; any other
    value is a target width."""
    if resolution in [1, 2, 4, 8]:
        return round(orig_w / (resolution_scale * resolution)), round(orig_h / (resolution_scale * resolution))
    if resolution == -1:
        global_down = orig_w / 1600 if orig_w > 1600 else 1
    else:
        global_down = orig_w / resolution
    scale = float(global_down) * float(resolution_scale)
    return int(orig_w / scale), int(orig_h / scale)


def _load_views(views, folder, resolution, device):
    from PIL import Image
    cams = []
    for v in views:
        img = Image.open(os.path.join(folder, os.path.basename(v["name"])))
        rgb = pil_to_torch(img, target_resolution(img.size[0], img.size[1], resolution))
        gt, mask = rgb[:3, ...], None
        if rgb.shape[1] == 4:            # sic: the reference tests dimension 1 (height), utils/camera_utils.py:45-46
            mask = rgb[3:4, ...]
        image = gt.clamp(0.0, 1.0).to(torch.float32)
        image = image * (mask if mask is not None else torch.ones((1, image.shape[1], image.shape[2])))   # cameras.py:43-46
        # (PILtoTorch permutes HWC -> CHW: the reference keeps that strided view, torch ops do not mind; the fused loss
        # kernels read plain (3, H, W) planes, so the frame is laid out once, here)
        cams.append(Camera(v["R"], v["T"], v["FovX"], v["FovY"], image.shape[2], image.shape[1], device=device,
                           image=image.contiguous().to(device)))
        cams[-1].image_name, cams[-1].uid = v["image_name"], len(cams) - 1
    return cams


def load_colmap_scene(path, images=None, gray=False, random=False, deblur=False, event=False, resolution=-1,
                      device="cpu"):
    """readColmapSceneInfo (scene/dataset_readers.py:134-227) + cameraList_from_camInfos (camera_utils.py:54-60)."""
    sp = os.path.join(path, "sparse/0")
    try:
        ex, intr = IO.read_images_binary(os.path.join(sp, "images.bin")), IO.read_cameras_binary(os.path.join(sp, "cameras.bin"))
    except Exception:
        ex, intr = IO.read_images_text(os.path.join(sp, "images.txt")), IO.read_cameras_text(os.path.join(sp, "cameras.txt"))
    views = IO.colmap_cameras_to_views(ex, intr)                     # sorted by image_name
    load = lambda d: _load_views(views, os.path.join(path, d), resolution, device)
    train = load("images" if images is None else images)
    blurry = load("images_blurry") if deblur else []
    events = load("images_event") if event else []
    test = load("renders")
    translate, radius = IO.nerf_normalization(views)
    ply_path = os.path.join(sp, "points3D.ply")
    if not os.path.exists(ply_path):
        try:
            xyz, rgb, _ = IO.read_points3D_binary(os.path.join(sp, "points3D.bin"))
        except Exception:
            xyz, rgb, _ = IO.read_points3D_text(os.path.join(sp, "points3D.txt"))
        IO.store_pointcloud_ply(ply_path, xyz, rgb)
    pcd = IO.fetch_pointcloud_ply(ply_path)
    pts, cols, nrm = np.array(pcd.points), np.array(pcd.colors), np.array(pcd.normals)
    if gray and not deblur:
        cols[:, :] = 0.5                                                 # dataset_readers.py:197-198
    if random:
        cols[:, :] = 0.5; nrm[:, :] = 0.5
        pts[:, :] = np.random.uniform(low=pts.min(0), high=pts.max(0), size=pts.shape)   # :201-216
    return SceneData(train, events, blurry, test, IO.BasicPointCloud(pts, cols, nrm), float(radius), translate, ply_path)


def create_from_pcd(pcd, spatial_lr_scale, dist2_fn, device="cuda"):
    """GaussianModel.create_from_pcd (scene/gaussian_model.py:124-147) -> pre-activation parameter dict."""
    from .synth import RGB2SH, inverse_sigmoid
    pts = torch.tensor(np.asarray(pcd.points)).float().to(device)
    col = RGB2SH(torch.tensor(np.asarray(pcd.colors)).float().to(device))
    N = pts.shape[0]
    feats = torch.zeros((N, 3, 16), dtype=torch.float32, device=device)
    feats[:, :3, 0] = col
    dist2 = torch.clamp_min(dist2_fn(pts), 0.0000001)
    scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
    rots = torch.zeros((N, 4), device=device); rots[:, 0] = 1
    opac = inverse_sigmoid(0.1 * torch.ones((N, 1), dtype=torch.float32, device=device))
    return dict(xyz=pts, features_dc=feats[:, :, 0:1].transpose(1, 2).contiguous(),
                features_rest=feats[:, :, 1:].transpose(1, 2).contiguous(), scaling=scales, rotation=rots, opacity=opac)


def evaluate_views(render_fn, test_cameras, index_list=(5, 25, 45, 65, 85)):
    """eval.py:118-152: render, clamp, convert render and GT to gray, mean SSIM / PSNR over the held-out views.
    `render_fn(camera) -> (3,H,W)`.  LPIPS needs downloaded network weights and is out of scope."""
    from . import losses
    ssim_t, psnr_t, per = 0.0, 0.0, []
    for index in index_list:
        cam = test_cameras[index]
        image = losses.rgb_to_grayscale(torch.clamp(render_fn(cam), 0.0, 1.0))
        gt = losses.rgb_to_grayscale(torch.clamp(cam.original_image.to(image.device), 0.0, 1.0))
        s, p = float(losses.ssim(image, gt)), float(losses.psnr(image, gt).mean())
        per.append((index, s, p)); ssim_t += s; psnr_t += p
    n = len(index_list)
    return {"ssim": ssim_t / n, "psnr": psnr_t / n, "per_view": per}
