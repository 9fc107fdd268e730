"""Mirror of the reference's boundary wrapper `render()` (gaussian_renderer/__init__.py:20-104) and
of the other two callers of the boundary, `render_depth()` (:106-189) and `render_point()` (:274-370, with its
`project_points` / `generate_depth_map` helpers :194-273), on the HIP operator -- same arguments, same results:

  * a zero `(N,3)` screen-space tensor that receives the NDC-unit mean gradients (`viewspace_points`, :28-32);
  * `pipe.compute_cov3D_python` -> covariance built in torch and passed as `cov3D_precomp` (:60-66);
  * colours: `override_color`, else SH evaluated in torch and clamped at 0 -- the reference FORCES this branch
    (`pipe.convert_SHs_python = True` at :71) -- or, when a caller clears the flag afterwards, SH inside the rasteriser;
  * returns {"render", "viewspace_points", "visibility_filter", "radii"} (:100-104).

The reference's own `render()` also runs unmodified on the drop-in packages; this module is for code
that lives on this repo's side of the boundary (EventTrainer users, tools) and wants the same call.  `GaussianView`
adapts pre-activation tensors to the `GaussianModel` getters `render()` reads (scene/gaussian_model.py:95-118,27-31).
"""
import math

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

# real spherical-harmonics constants (utils/sh_utils.py:24-55)
_C0 = 0.28209479177387814
_C1 = 0.4886025119029199
_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435)
_C4 = (2.5033429417967046, -1.7701307697799304, 0.9461746957575601, -0.6690465435572892, 0.10578554691520431,
       -0.6690465435572892, 0.47308734787878004, -1.7701307697799304, 0.6258357354491761)


def eval_sh(deg, sh, dirs):
    """utils/sh_utils.py:57-112 for deg 0..4: sh (..., C, (max_deg+1)^2), dirs (..., 3) unit vectors -> (..., C)."""
    if not 0 <= deg <= 4:
        raise ValueError("SH degree must be 0..4")
    if sh.shape[-1] < (deg + 1) ** 2:
        raise ValueError("not enough SH coefficients for this degree")
    res = _C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        res = res - _C1 * y * sh[..., 1] + _C1 * z * sh[..., 2] - _C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + _C2[0] * xy * sh[..., 4] + _C2[1] * yz * sh[..., 5] + _C2[2] * (2.0 * zz - xx - yy) * sh[..., 6]
                   + _C2[3] * xz * sh[..., 7] + _C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                res = (res + _C3[0] * y * (3 * xx - yy) * sh[..., 9] + _C3[1] * xy * z * sh[..., 10]
                       + _C3[2] * y * (4 * zz - xx - yy) * sh[..., 11]
                       + _C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12]
                       + _C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + _C3[5] * z * (xx - yy) * sh[..., 14]
                       + _C3[6] * x * (xx - 3 * yy) * sh[..., 15])
                if deg > 3:                                      # utils/sh_utils.py:97-110
                    res = (res + _C4[0] * xy * (xx - yy) * sh[..., 16] + _C4[1] * yz * (3 * xx - yy) * sh[..., 17]
                           + _C4[2] * xy * (7 * zz - 1) * sh[..., 18] + _C4[3] * yz * (7 * zz - 3) * sh[..., 19]
                           + _C4[4] * (zz * (35 * zz - 30) + 3) * sh[..., 20] + _C4[5] * xz * (7 * zz - 3) * sh[..., 21]
                           + _C4[6] * (xx - yy) * (7 * zz - 1) * sh[..., 22] + _C4[7] * xz * (xx - 3 * yy) * sh[..., 23]
                           + _C4[8] * (xx * (xx - 3 * yy) - yy * (3 * xx - yy)) * sh[..., 24])
    return res


def build_covariance(scaling, rotation, scaling_modifier=1.0):
    """scene/gaussian_model.py:27-31 (+ utils/general_utils.py:64-110): (N,6) upper triangle of R S S^T R^T with the
    quaternion normalised first (build_rotation does)."""
    q = rotation / rotation.norm(dim=1, keepdim=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)), dim=1).reshape(-1, 3, 3)
    L = R * (scaling_modifier * scaling)[:, None, :]
    S = L @ L.transpose(1, 2)
    return torch.stack((S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]), dim=1)


class GaussianView:
    """The `GaussianModel` getters that render() reads, over pre-activation tensors (a dict as synth.make_scene /
    scene_io.create_from_pcd return, or EventTrainer.export_groups()-style [param, ...] lists)."""

    def __init__(self, params, active_sh_degree=3, max_sh_degree=3):
        first = lambda v: v[0] if isinstance(v, (list, tuple)) else v
        g = {k: first(v) for k, v in params.items()}
        self._xyz, self._scaling, self._rotation, self._opacity = g["xyz"], g["scaling"], g["rotation"], g["opacity"]
        self._features_dc = g["features_dc"] if "features_dc" in g else g["f_dc"]
        self._features_rest = g["features_rest"] if "features_rest" in g else g["f_rest"]
        self.active_sh_degree, self.max_sh_degree = int(active_sh_degree), int(max_sh_degree)

    get_xyz = property(lambda self: self._xyz)
    get_scaling = property(lambda self: torch.exp(self._scaling))
    get_rotation = property(lambda self: torch.nn.functional.normalize(self._rotation))
    get_opacity = property(lambda self: torch.sigmoid(self._opacity))
    get_features = property(lambda self: torch.cat((self._features_dc, self._features_rest), dim=1))

    def get_covariance(self, scaling_modifier=1.0):
        return build_covariance(self.get_scaling, self._rotation, scaling_modifier)


class PipelineParams:
    """arguments/__init__.py:57-62"""

    def __init__(self, convert_SHs_python=False, compute_cov3D_python=False, debug=False):
        self.convert_SHs_python, self.compute_cov3D_python, self.debug = convert_SHs_python, compute_cov3D_python, debug


def _rasterize(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, colors_precomp, shs):
    screenspace_points = torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:           # noqa: BLE001 -- as the reference: a no-grad context has nothing to retain
        pass
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=pipe.debug)
    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales, rotations = pc.get_scaling, pc.get_rotation
    rendered_image, radii = GaussianRasterizer(raster_settings=raster_settings)(
        means3D=pc.get_xyz, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp,
        opacities=pc.get_opacity, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii}


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, force_python_sh=True):
    """gaussian_renderer/__init__.py:20-104.  `force_python_sh=True` reproduces the reference's line :71, which sets
    pipe.convert_SHs_python = True on every call; pass False to honour the flag as upstream 3DGS does."""
    if force_python_sh:
        pipe.convert_SHs_python = True
    shs = colors_precomp = None
    if override_color is not None:
        colors_precomp = override_color
    elif pipe.convert_SHs_python:
        feats = pc.get_features
        shs_view = feats.transpose(1, 2).reshape(-1, 3, (pc.max_sh_degree + 1) ** 2)
        dir_pp = pc.get_xyz - viewpoint_camera.camera_center.repeat(feats.shape[0], 1)
        dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
        colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp_normalized) + 0.5, 0.0)
    else:
        shs = pc.get_features
    return _rasterize(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, colors_precomp, shs)


def render_depth(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
    """gaussian_renderer/__init__.py:106-189: the colour of every Gaussian is its distance to the camera centre
    (+0.5, clamped at 0, on all three channels), composited like any other colour."""
    distance = (pc.get_xyz - viewpoint_camera.camera_center.repeat(pc.get_xyz.shape[0], 1)).norm(dim=1, keepdim=True)
    colors_precomp = torch.clamp_min(distance.repeat(1, 3) + 0.5, 0.0)
    return _rasterize(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, colors_precomp, None)


def project_points(points_3d, full_proj_transform, epsilon=1e-4):
    """gaussian_renderer/__init__.py:194-220: (N,3) world points -> (N,2) NDC through the row-vector full projection,
    normalised by (w + 1e-4) -- the reference's own restatement of the rasteriser's projection (its eps is 1e-4)."""
    ones = torch.ones(points_3d.shape[0], 1, dtype=points_3d.dtype, device=points_3d.device)
    clip = torch.cat((points_3d, ones), dim=1) @ full_proj_transform.to(points_3d.dtype)
    return clip[:, :2] / (clip[:, 3:4] + epsilon)


def generate_depth_map(points_3d, camera_center, projection_matrix, image_size, radii=None):
    """gaussian_renderer/__init__.py:221-273: nearest-point depth map -- every point goes to the pixel its centre falls in
    (ndc2Pix, then int() truncation) and a pixel keeps the smallest distance to the camera centre; pixels no point hits
    hold +inf.  The reference walks the points in a Python loop on the host; this is one scatter-min on the points'
    device.  `radii` is accepted and unused, as upstream."""
    with torch.no_grad():
        width, height = int(image_size[0]), int(image_size[1])
        pts = points_3d.detach()
        dist = (pts - camera_center.to(pts.device)).norm(dim=1)
        ndc = project_points(pts, projection_matrix.to(pts.device))
        x = (((ndc[:, 0] + 1) * width - 1) * 0.5).to(torch.int64)        # .to(int64) truncates toward zero, like int()
        y = (((ndc[:, 1] + 1) * height - 1) * 0.5).to(torch.int64)
        ok = (x >= 0) & (x < width) & (y >= 0) & (y < height)
        depth = torch.full((height * width,), float("inf"), dtype=torch.float32, device=pts.device)
        depth.scatter_reduce_(0, (y * width + x)[ok], dist[ok].to(torch.float32), reduce="amin", include_self=True)
        return depth.view(height, width)


def render_point(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
    """gaussian_renderer/__init__.py:274-370 (render.py:305): one rasteriser call for the radii, then the nearest-point
    depth map of the visible Gaussians (radius > 0) whose opacity exceeds 0.8."""
    out = render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color, force_python_sh=False)
    with torch.no_grad():
        keep = (out["radii"] > 0) & (pc.get_opacity.reshape(-1) > 0.8)
        points = pc.get_xyz[keep]
    return generate_depth_map(points, viewpoint_camera.camera_center, viewpoint_camera.full_proj_transform,
                              [int(viewpoint_camera.image_width), int(viewpoint_camera.image_height)], out["radii"][keep])
