"""The extension-level surface of upstream `diff_gaussian_rasterization._C` ("C++/HIP extension ABI"):
the three functions the upstream Python layer calls, with its positional signatures, its "empty tensor = not provided"
convention and its return tuples -- for code that keeps upstream's own `__init__.py` and swaps only the extension.
Implementation: event_3dgs_amd/rasterizer.py over the C ABI (include/e3dgs_hip.h)."""
import torch

from event_3dgs_amd import rasterizer as _r

# The compiled module (csrc/ext.cpp, built by build_ext.py / __graft_entry__.build()) exports the same three functions;
# when it is present they ARE this module's functions, otherwise the ctypes implementations below serve.
_native = _r.native_ext()


def _opt(t):
    return None if t is None or t.numel() == 0 else t


def rasterize_gaussians(bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                        projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered,
                        debug):
    """-> (num_rendered, out_color (3,H,W), radii (P,) int32, geomBuffer, binningBuffer, imgBuffer) (uint8 tensors)."""
    rs = _r.GaussianRasterizationSettings(int(image_height), int(image_width), float(tan_fovx), float(tan_fovy), bg,
                                          float(scale_modifier), viewmatrix, projmatrix, int(degree), campos,
                                          bool(prefiltered), bool(debug))
    raw = _r.forward_raw(means3D, _opt(sh), _opt(colors), opacity, _opt(scales), _opt(rotations), _opt(cov3D_precomp), rs)
    return raw["num_rendered"], raw["color"], raw["radii"], raw["geom"], raw["binning"], raw["image"]


def rasterize_gaussians_backward(bg, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos,
                                 geomBuffer, num_rendered, binningBuffer, imgBuffer, debug):
    """-> (dL_dmeans2D (P,3), dL_dcolors (P,3), dL_dopacity (P,1), dL_dmeans3D (P,3), dL_dcov3D (P,6),
    dL_dsh (P,M,3), dL_dscales (P,3), dL_drotations (P,4)); groups that do not apply come back as zeros."""
    P = means3D.shape[0]
    dev = means3D.device
    H, W = int(dL_dout_color.shape[-2]), int(dL_dout_color.shape[-1])
    sh_c, colors_c, scales_c, rots_c, cov_c = (_r._prep(_opt(t), n) for t, n in (
        (sh, "sh"), (colors, "colors"), (scales, "scales"), (rotations, "rotations"), (cov3D_precomp, "cov3D_precomp")))
    M = 0 if sh_c is None else sh_c.shape[1]
    rs = _r.GaussianRasterizationSettings(H, W, float(tan_fovx), float(tan_fovy), bg, float(scale_modifier), viewmatrix,
                                          projmatrix, int(degree), campos, False, bool(debug))
    z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
    out = dict(means2D=z(P, 3), opacities=z(P, 1), means3D=z(P, 3), colors=z(P, 3) if colors_c is not None else None,
               cov3D=z(P, 6) if cov_c is not None else None, sh=z(P, M, 3) if sh_c is not None else None,
               scales=z(P, 3) if cov_c is None else None, rots=z(P, 4) if cov_c is None else None)
    raw = dict(num_rendered=int(num_rendered), M=M, settings=rs, flags=0,
               inputs=(_r._prep(means3D, "means3D"), sh_c, colors_c, scales_c, rots_c, cov_c), opacities=None,
               consts=tuple(_r._prep(t, n) for t, n in ((bg, "bg"), (viewmatrix, "viewmatrix"),
                                                         (projmatrix, "projmatrix"), (campos, "campos"))),
               radii=radii, geom=geomBuffer, binning=binningBuffer, image=imgBuffer)
    _r.backward_raw(raw, dL_dout_color, out)
    fill = lambda t, *shape: t if t is not None else z(*shape)
    return (out["means2D"], fill(out["colors"], P, 3), out["opacities"], out["means3D"], fill(out["cov3D"], P, 6),
            fill(out["sh"], P, max(M, 0), 3), fill(out["scales"], P, 3), fill(out["rots"], P, 4))


def mark_visible(means3D, viewmatrix, projmatrix):
    """-> bool (P,): view-space z > 0.2 (upstream markVisible / in_frustum)."""
    pos = _r._prep(means3D, "means3D")
    if pos is None:
        return torch.zeros(0, dtype=torch.bool, device=means3D.device)
    return _r._mark_visible(pos, _r._prep(viewmatrix, "viewmatrix"), _r._prep(projmatrix, "projmatrix")).bool()


if _native is not None:
    rasterize_gaussians = _native.rasterize_gaussians
    rasterize_gaussians_backward = _native.rasterize_gaussians_backward
    mark_visible = _native.mark_visible
