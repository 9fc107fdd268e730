"""Per-view matrices consumed by the rasteriser, in the reference's conventions.

The fields of scene/cameras.py:17-57 (Camera / MiniCam) with the matrices of utils/graphics_utils.py:38-49
(getWorld2View2) and :51-71 (getProjectionMatrix) in closed form.  Pinned by tests/golden/cameras.npz.
"""
import math

import numpy as np
import torch


def getWorld2View2(R, t, translate=np.array([0.0, 0.0, 0.0]), scale=1.0):
    """World-to-view matrix of utils/graphics_utils.py:38-49 in closed form.  R is the camera-to-world rotation (stored
    transposed, scene/dataset_readers.py:246) and t the world-to-camera translation, so the camera sits at c = -R t; the
    reference's recentring moves it to (c + translate) * scale and keeps the orientation: x_view = R^T (x - c')."""
    R = np.asarray(R, np.float64)
    centre = (-R @ np.asarray(t, np.float64) + np.asarray(translate, np.float64)) * scale
    w2v = np.eye(4)
    w2v[:3, :3] = R.T
    w2v[:3, 3] = -R.T @ centre
    return w2v.astype(np.float32)


def getProjectionMatrix(znear, zfar, fovX, fovY):
    """The symmetric-frustum projection of utils/graphics_utils.py:51-71: x, y scaled by the cotangents of the half
    angles (the frustum is centred, so there is no shear term), z mapped to [0, 1] over [znear, zfar], w_clip = z_view."""
    P = torch.zeros(4, 4)
    P[0, 0] = 1.0 / math.tan(0.5 * fovX)
    P[1, 1] = 1.0 / math.tan(0.5 * fovY)
    depth = zfar - znear
    P[2, 2] = zfar / depth
    P[2, 3] = -(zfar * znear) / depth
    P[3, 2] = 1.0
    return P


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


class Camera:
    """The fields of scene/cameras.py:Camera that render() reads (gaussian_renderer/__init__.py:35-48)."""

    def __init__(self, R, T, FoVx, FoVy, width, height, device="cpu", znear=0.01, zfar=100.0, image=None,
                 trans=np.array([0.0, 0.0, 0.0]), scale=1.0):
        self.R, self.T, self.FoVx, self.FoVy = R, T, FoVx, FoVy
        self.image_width, self.image_height = int(width), int(height)
        self.znear, self.zfar = znear, zfar
        self.original_image = image
        wvt = torch.tensor(getWorld2View2(R, T, trans, scale)).transpose(0, 1)
        proj = getProjectionMatrix(znear, zfar, FoVx, FoVy).transpose(0, 1)
        full = wvt.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0)
        center = wvt.inverse()[3, :3]
        # the reference leaves world_view_transform / camera_center as strided views;
        # they are kept strided here too so the op's .contiguous() handling is exercised
        self.world_view_transform = wvt.to(device)
        self.projection_matrix = proj.to(device)
        self.full_proj_transform = full.to(device)
        self.camera_center = center.to(device)


def look_at(eye, target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)):
    """Returns (R, T) in the reference's storage convention for a camera at `eye` looking at `target`
    (camera +z forward, +y down)."""
    eye, target, up = (np.asarray(v, np.float64) for v in (eye, target, up))
    fwd = target - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    Rw2c = np.stack([right, down, fwd], 0)
    return Rw2c.T.copy(), -Rw2c @ eye


def orbit_camera(k, K, width, height, device="cpu", radius=4.0, elevation=0.5, az0=0.7, daz=0.0,
                 fovx=0.6911112070083618):
    """Camera rig of SURVEY 8(d): look-at origin from `radius`, azimuth az0 + 2*pi*k/K (+daz)."""
    az = az0 + 2.0 * math.pi * k / K + daz
    eye = radius * np.array([math.cos(elevation) * math.cos(az), math.cos(elevation) * math.sin(az),
                             math.sin(elevation)])
    R, T = look_at(eye)
    fovy = 2.0 * math.atan(math.tan(fovx / 2.0) * height / width)
    return Camera(R, T, fovx, fovy, width, height, device=device)
