"""Host-side mirror of the reference's rasteriser operator API, bound to the HIP C ABI.

Mirrors the Python surface of the un-vendored `diff_gaussian_rasterization` package that the
reference imports at gaussian_renderer/__init__.py:15 and drives at :38-53,89-97:

    GaussianRasterizationSettings  12-field NamedTuple (same keyword set as :38-51)
    GaussianRasterizer             nn.Module; forward(means3D, means2D, opacities, shs=None,
                                   colors_precomp=None, scales=None, rotations=None,
                                   cov3D_precomp=None) -> (color (3,H,W), radii (P,) int32)
    rasterize_gaussians / _RasterizeGaussians   autograd.Function keeping the three scratch
                                   buffers alive between forward and backward.

Gradient tuple order: means3D, means2D, sh, colors_precomp, opacities,
scales, rotations, cov3D_precomp, raster_settings(None).
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


_NATIVE = None


def native_ext():
    """The compiled torch extension diff_gaussian_rasterization._C_native (csrc/ext.cpp: torch::Tensor <-> C ABI in C++), or
    None when it has not been built or E3DGS_NATIVE_EXT=0.  The autograd operator goes through it when present -- the
    tensor marshalling of one forward + backward costs ~20 us in C++ against ~120 us through ctypes; results are
    identical (same C ABI underneath)."""
    global _NATIVE
    if _NATIVE is None:
        import os
        _NATIVE = False
        if os.environ.get("E3DGS_NATIVE_EXT", "1") != "0":
            try:
                _lib.lib()                                  # loud failure if the HIP library itself is missing
                from diff_gaussian_rasterization import _C_native
                if _C_native.abi_version() == _lib.ABI_VERSION:
                    _NATIVE = _C_native
            except ImportError:
                _NATIVE = False
    return _NATIVE or None


def _prep(t, name, device=None):
    """fp32, contiguous, on the GPU.  Inputs may arrive strided (scene/cameras.py:54-57)."""
    if t is None or t.numel() == 0:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA/HIP tensor (this op has no CPU path)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


class ScratchPool:
    """Persistent, geometrically growing device buffers keyed by name.  A training loop that re-renders every
    iteration asks for slightly different scratch sizes each time (the instance count moves with the scene);
    going through the allocator for every size means hipMalloc / hipFree traffic whose latency is at the mercy
    of the driver.  A pool buffer is reused as long as it is large enough and grows by 25 % when it is not.
    Buffers handed out are only valid until the next request of the same name (one iteration in flight)."""

    GROWTH = 1.25

    def __init__(self, device):
        self.device = device
        self._bufs = {}

    def get(self, name, nbytes):
        nbytes = int(nbytes)
        buf = self._bufs.get(name)
        if buf is None or buf.numel() < nbytes:
            buf = None
            self._bufs.pop(name, None)           # release the old block before asking for the bigger one
            buf = torch.empty(max(int(nbytes * self.GROWTH), 256), dtype=torch.uint8, device=self.device)
            self._bufs[name] = buf
        return buf

    def typed(self, name, shape, dtype=torch.float32):
        n = 1
        for d in shape:
            n *= int(d)
        item = torch.empty(0, dtype=dtype).element_size()
        return self.get(name, max(n * item, 1))[:n * item].view(dtype).view(*shape)

    def clear(self):
        self._bufs.clear()


class _Scratch:
    """One growable uint8 torch tensor handed to the C ABI through an allocation callback.  The callback closes over a
    holder list, not over the object: a ctypes callback that references its owner forms a cycle (object -> callback ->
    bound method -> object) that only the cyclic garbage collector frees -- with ~1 GB of scratch per forward behind it, a
    loop that keeps the collector out of its timing (bench.py, timeit) would allocate fresh device memory every call."""

    def __init__(self, device, pool=None, name=None):
        self.device = device
        self.pool, self.name = pool, name
        holder = self._holder = [torch.empty(0, dtype=torch.uint8, device=device)]

        def alloc(_user, nbytes):
            if pool is not None:
                holder[0] = pool.get(name, nbytes)
            else:
                holder[0] = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            return holder[0].data_ptr()
        self.cb = _lib.ALLOC_FN(alloc)

    @property
    def tensor(self):
        return self._holder[0]


def _mark_visible(pos, view, proj):
    """uint8 (P,) on the device: 1 where view-space z > 0.2 (e3dgs_mark_visible); inputs already contiguous fp32."""
    P = pos.shape[0]
    present = torch.zeros(P, dtype=torch.uint8, device=pos.device)
    if P:
        with torch.cuda.device(pos.device):
            rc = _lib.lib().e3dgs_mark_visible(P, _lib.ptr(pos), _lib.ptr(view), _lib.ptr(proj), _lib.ptr(present),
                                               _lib.current_stream())
        _lib.check(rc, "e3dgs_mark_visible")
    return present


def _cpu_snapshot(args):
    """Host copies of an argument tuple (tensors cloned to the CPU, everything else kept)."""
    return tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


def _debug_guard(enabled, args, path, what, fn):
    """debug=True contract of the operator ([UPSTREAM] _RasterizeGaussians, SURVEY 8b "Errors"): keep host copies of
    the arguments, and if the call raises, write them to `path` with torch.save before re-raising."""
    if not enabled:
        return fn()
    snapshot = _cpu_snapshot(args)
    try:
        return fn()
    except Exception:
        torch.save(snapshot, path)
        print("\nAn error occured in %s. Please forward %s for debugging." % (what, path))
        raise


def _check_operator_args(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp):
    """The argument rules of upstream's operator, with its messages (SURVEY 8b "Errors")."""
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a CUDA/HIP tensor (this op has no CPU path)")
    P = means3D.shape[0]
    if P:
        if opacities is None or opacities.numel() != P:
            raise RuntimeError("opacities must have P elements")
        if (sh is None or sh.numel() == 0) == (colors_precomp is None or colors_precomp.numel() == 0):
            raise RuntimeError("Please provide excatly one of either SHs or precomputed colors!")
        has_sr = scales is not None and rotations is not None and scales.numel() and rotations.numel()
        if bool(has_sr) == (cov3Ds_precomp is not None and cov3Ds_precomp.numel() != 0):
            raise RuntimeError("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")


def cpp_autograd_ext():
    """The compiled extension when its C++ autograd node may serve the operator (csrc/ext.cpp RasterizeFunction;
    E3DGS_CPP_AUTOGRAD=0: the Python autograd.Function below over the same extension functions)."""
    import os
    ext = native_ext()
    if ext is None or not hasattr(ext, "rasterize_autograd") or os.environ.get("E3DGS_CPP_AUTOGRAD", "1") == "0":
        return None
    return ext


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, flags=0):
    """The operator.  With the compiled extension the autograd node itself lives in C++ (no Python frame per render and
    per backward); debug / prefiltered calls -- which snapshot their arguments or check the caller's promise on the host
    -- and builds without the extension take the Python autograd.Function.  Identical results (same C ABI underneath).
    flags: the C ABI's flags word (E3DGS_FLAG_PREACT: raw log-scales / quaternions / logits, activations in the kernels)."""
    rs = raster_settings
    ext = cpp_autograd_ext()
    if ext is None or rs.debug or rs.prefiltered:
        if flags:
            raise RuntimeError("flags need the compiled extension's autograd node")
        return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                         cov3Ds_precomp, raster_settings)
    _check_operator_args(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
    empty = means3D.new_empty(0)
    e = lambda t: empty if t is None else t
    return ext.rasterize_autograd(means3D, e(means2D), e(sh), e(colors_precomp), e(opacities), e(scales), e(rotations),
                                  e(cov3Ds_precomp), rs.bg, rs.viewmatrix, rs.projmatrix, rs.campos,
                                  float(rs.scale_modifier), float(rs.tanfovx), float(rs.tanfovy), int(rs.image_height),
                                  int(rs.image_width), int(rs.sh_degree), int(flags))


def forward_raw(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, flags=0):
    """One call of e3dgs_rasterize_forward.  Returns a dict with the outputs, the prepared
    (contiguous) inputs and the three scratch tensors that backward needs."""
    import ctypes as C
    L = _lib.lib()
    rs = raster_settings
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    dev = means3D.device
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a CUDA/HIP tensor (this op has no CPU path)")
    P = means3D.shape[0]
    H, W = int(rs.image_height), int(rs.image_width)
    means3D_c = _prep(means3D, "means3D")
    sh_c = _prep(sh, "shs")
    colors_c = _prep(colors_precomp, "colors_precomp")
    opac_c = _prep(opacities, "opacities")
    scales_c = _prep(scales, "scales")
    rots_c = _prep(rotations, "rotations")
    cov_c = _prep(cov3Ds_precomp, "cov3D_precomp")
    bg = _prep(rs.bg, "bg"); view = _prep(rs.viewmatrix, "viewmatrix")
    proj = _prep(rs.projmatrix, "projmatrix"); campos = _prep(rs.campos, "campos")
    M = 0 if sh_c is None else (sh_c.shape[0] // 3 if (flags & _lib.FLAG_SH_PLANAR) else sh_c.shape[1])
    if P:
        if opac_c is None or opac_c.numel() != P:
            raise RuntimeError("opacities must have P elements")
        if (sh_c is None) == (colors_c is None):
            raise RuntimeError("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales_c is None or rots_c is None) == (cov_c is None)):
            raise RuntimeError("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    if rs.prefiltered and P:
        # The caller promises that every Gaussian passes the near-plane test; upstream's kernel traps the device on a
        # violation ([UPSTREAM] in_frustum).  Here the same test (e3dgs_mark_visible) runs first
        # and a violation is a Python exception instead of a lost context.
        if not bool(_mark_visible(means3D_c, view, proj).all()):
            raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
    out_color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
    radii = torch.empty(P, dtype=torch.int32, device=dev)
    geom, binning, img = _Scratch(dev), _Scratch(dev), _Scratch(dev)
    num_rendered = C.c_int(0)
    with torch.cuda.device(dev):
        rc = L.e3dgs_rasterize_forward(
            geom.cb, None, binning.cb, None, img.cb, None, P, int(rs.sh_degree), M, _lib.ptr(bg), W, H,
            _lib.ptr(means3D_c), _lib.ptr(sh_c), _lib.ptr(colors_c), _lib.ptr(opac_c), _lib.ptr(scales_c),
            float(rs.scale_modifier), _lib.ptr(rots_c), _lib.ptr(cov_c), _lib.ptr(view), _lib.ptr(proj),
            _lib.ptr(campos), float(rs.tanfovx), float(rs.tanfovy), int(bool(rs.prefiltered)),
            _lib.ptr(out_color), _lib.ptr(radii), int(bool(rs.debug)), int(flags), C.byref(num_rendered),
            _lib.current_stream())
    _lib.check(rc, "e3dgs_rasterize_forward")
    return dict(color=out_color, radii=radii, num_rendered=num_rendered.value, M=M, settings=rs, flags=int(flags),
                inputs=(means3D_c, sh_c, colors_c, scales_c, rots_c, cov_c), opacities=opac_c,
                consts=(bg, view, proj, campos),
                geom=geom.tensor, binning=binning.tensor, image=img.tensor)


class PendingForward:
    """Result of forward_begin(); finish() it after the stream has been synchronised."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def forward_begin(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, flags=0,
                  count_host=None):
    """Enqueue-only first half of the forward (e3dgs_rasterize_forward_begin).  `count_host` is a pinned
    int32 host tensor with one element that receives the instance count asynchronously."""
    L = _lib.lib()
    rs = raster_settings
    dev = means3D.device
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a CUDA/HIP tensor (this op has no CPU path)")
    P = means3D.shape[0]
    H, W = int(rs.image_height), int(rs.image_width)
    means3D_c = _prep(means3D, "means3D")
    sh_c, colors_c, opac_c = _prep(sh, "shs"), _prep(colors_precomp, "colors_precomp"), _prep(opacities, "opacities")
    scales_c, rots_c, cov_c = _prep(scales, "scales"), _prep(rotations, "rotations"), _prep(cov3Ds_precomp, "cov3D_precomp")
    bg = _prep(rs.bg, "bg"); view = _prep(rs.viewmatrix, "viewmatrix")
    proj = _prep(rs.projmatrix, "projmatrix"); campos = _prep(rs.campos, "campos")
    M = 0 if sh_c is None else (sh_c.shape[0] // 3 if (flags & _lib.FLAG_SH_PLANAR) else sh_c.shape[1])
    if count_host is None:
        count_host = torch.zeros(1, dtype=torch.int32).pin_memory()
    radii = torch.empty(P, dtype=torch.int32, device=dev)
    geom, img = _Scratch(dev), _Scratch(dev)
    with torch.cuda.device(dev):
        rc = L.e3dgs_rasterize_forward_begin(
            geom.cb, None, img.cb, None, P, int(rs.sh_degree), M, W, H, _lib.ptr(means3D_c), _lib.ptr(sh_c),
            _lib.ptr(colors_c), _lib.ptr(opac_c), _lib.ptr(scales_c), float(rs.scale_modifier), _lib.ptr(rots_c),
            _lib.ptr(cov_c), _lib.ptr(view), _lib.ptr(proj), _lib.ptr(campos), float(rs.tanfovx), float(rs.tanfovy),
            _lib.ptr(radii), int(bool(rs.debug)), int(flags), count_host.data_ptr(), _lib.current_stream())
    _lib.check(rc, "e3dgs_rasterize_forward_begin")
    return PendingForward(rs=rs, flags=int(flags), P=P, W=W, H=H, M=M, radii=radii, geom=geom.tensor, image=img.tensor,
                          inputs=(means3D_c, sh_c, colors_c, scales_c, rots_c, cov_c), opacities=opac_c,
                          consts=(bg, view, proj, campos), count_host=count_host, device=dev)


def forward_finish(pending):
    """Second half; the caller has synchronised the stream since forward_begin()."""
    L = _lib.lib()
    p = pending
    I = int(p.count_host[0])
    out_color = torch.empty(3, p.H, p.W, dtype=torch.float32, device=p.device)
    binning = _Scratch(p.device)
    with torch.cuda.device(p.device):
        rc = L.e3dgs_rasterize_forward_finish(binning.cb, None, p.P, p.W, p.H, _lib.ptr(p.consts[0]), _lib.ptr(p.geom),
                                              _lib.ptr(p.image), I, _lib.ptr(out_color), int(bool(p.rs.debug)),
                                              p.flags & _lib.OPTION_MASK, _lib.current_stream())
    _lib.check(rc, "e3dgs_rasterize_forward_finish")
    return dict(color=out_color, radii=p.radii, num_rendered=I, M=p.M, settings=p.rs, flags=p.flags, inputs=p.inputs,
                opacities=p.opacities, consts=p.consts, geom=p.geom, binning=binning.tensor, image=p.image)


def state_views(raw, P, W, H):
    """Typed views into the scratch buffers of a forward_raw() result (tests / tools)."""
    import ctypes as C
    offs = (C.c_size_t * 9)()
    I = raw["num_rendered"]
    _lib.lib().e3dgs_state_offsets(P, I, W, H, offs)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    def view(buf, off, dtype, count, shape):
        nbytes = count * torch.empty(0, dtype=dtype).element_size()
        return buf[off:off + nbytes].view(dtype).reshape(shape)
    g, b, im = raw["geom"], raw["binning"], raw["image"]
    rec = view(g, offs[0], torch.float32, 12 * P, (P, 12))
    offs_b = (C.c_size_t * 2)()
    _lib.lib().e3dgs_state_offsets_binning(I, offs_b)
    return dict(
        strip_mask=view(b, offs_b[0], torch.uint8, I, (I,)), touched=view(b, offs_b[1], torch.uint8, I, (I,)),
        recA=rec[:, 0:4], recB=rec[:, 4:8], recC=rec[:, 8:10], clamped=view(g, offs[3], torch.int32, P, (P,)),
        rect=view(g, offs[4], torch.int32, 2 * P, (P, 2)),
        perm=view(b, offs[5], torch.int32, I, (I,)),
        emit_gid=view(b, _lib.lib().e3dgs_state_offset_emit_gid(I), torch.int32, I, (I,)),
        # sorted Gaussian ids (what the reference calls point_list): emit_gid gathered through perm
        point_list=view(b, _lib.lib().e3dgs_state_offset_emit_gid(I), torch.int32, I, (I,))[
            view(b, offs[5], torch.int32, I, (I,)).long()],
        ranges=view(im, offs[6], torch.int32, 2 * gx * gy, (gx * gy, 2)),
        final_T=view(im, offs[7], torch.float32, W * H, (H, W)),
        n_contrib=view(im, offs[8], torch.int32, W * H, (H, W)))


def state_views_multi(raw, P, W, H):
    """Typed views into the image scratch of a forward_multi() result: ranges (n*tiles, 2), final_T / n_contrib (n, H, W)."""
    import ctypes as C
    n = len(raw["settings_list"])
    offs = (C.c_size_t * 9)()
    _lib.lib().e3dgs_state_offsets_multi(n, P, int(raw["num_rendered"]), W, H, offs)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    im = raw["image"]
    def view(off, dtype, count, shape):
        nbytes = count * torch.empty(0, dtype=dtype).element_size()
        return im[off:off + nbytes].view(dtype).reshape(shape)
    return dict(ranges=view(offs[6], torch.int32, 2 * gx * gy * n, (gx * gy * n, 2)),
                final_T=view(offs[7], torch.float32, W * H * n, (n, H, W)),
                n_contrib=view(offs[8], torch.int32, W * H * n, (n, H, W)))


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        rs = raster_settings
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
                rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
        ext = native_ext()
        if ext is not None:
            return _RasterizeGaussians._forward_native(ctx, ext, args, means3D, sh, colors_precomp, opacities, scales,
                                                       rotations, cov3Ds_precomp, rs)
        raw = _debug_guard(rs.debug, args, "snapshot_fw.dump", "forward", lambda: forward_raw(
            means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings))
        ctx.raster_settings = raster_settings
        ctx.num_rendered = raw["num_rendered"]
        ctx.consts = raw["consts"]
        ctx.M = raw["M"]
        ctx.native = False
        ctx.save_for_backward(*raw["inputs"], raw["radii"], raw["geom"], raw["binning"], raw["image"])
        ctx.mark_non_differentiable(raw["radii"])
        return raw["color"], raw["radii"]

    @staticmethod
    def _forward_native(ctx, ext, args, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs):
        """Upstream's own Python layer, verbatim in structure: empty tensor = not provided, the extension returns
        (num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer)."""
        _check_operator_args(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
        P = means3D.shape[0]
        empty = means3D.new_empty(0)
        e = lambda t: empty if t is None else t
        def call():
            # (the caller's promise is checked first: upstream's kernel traps the device on a violation, see forward_raw)
            if rs.prefiltered and P and not bool(ext.mark_visible(means3D, rs.viewmatrix, rs.projmatrix).all()):
                raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
            return ext.rasterize_gaussians(
                rs.bg, means3D, e(colors_precomp), e(opacities), e(scales), e(rotations), float(rs.scale_modifier),
                e(cov3Ds_precomp), rs.viewmatrix, rs.projmatrix, float(rs.tanfovx), float(rs.tanfovy),
                int(rs.image_height), int(rs.image_width), e(sh), int(rs.sh_degree), rs.campos, bool(rs.prefiltered),
                bool(rs.debug))
        num_rendered, color, radii, geomB, binB, imgB = _debug_guard(rs.debug, args, "snapshot_fw.dump", "forward", call)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.native = True
        ctx.save_for_backward(means3D, e(sh), e(colors_precomp), e(scales), e(rotations), e(cov3Ds_precomp), radii, geomB,
                              binB, imgB)
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii):
        rs = ctx.raster_settings
        means3D, sh, colors, scales, rots, cov, radii, geomB, binB, imgB = ctx.saved_tensors
        if ctx.native:
            ext = native_ext()
            g = grad_out_color if grad_out_color.dtype == torch.float32 else grad_out_color.float()
            args = (rs.bg, means3D, radii, colors, scales, rots, rs.scale_modifier, cov, rs.viewmatrix, rs.projmatrix,
                    rs.tanfovx, rs.tanfovy, g, sh, rs.sh_degree, rs.campos, geomB, ctx.num_rendered, binB, imgB, rs.debug)
            (d_means2D, d_colors, d_opac, d_means3D, d_cov, d_sh, d_scales, d_rots) = _debug_guard(
                rs.debug, args, "snapshot_bw.dump", "backward", lambda: ext.rasterize_gaussians_backward(
                    rs.bg, means3D, radii, colors, scales, rots, float(rs.scale_modifier), cov, rs.viewmatrix,
                    rs.projmatrix, float(rs.tanfovx), float(rs.tanfovy), g, sh, int(rs.sh_degree), rs.campos, geomB,
                    int(ctx.num_rendered), binB, imgB, bool(rs.debug)))
            n = lambda t, inp: t if inp.numel() else None
            has_cov = cov.numel() != 0
            return (d_means3D, d_means2D, n(d_sh, sh), n(d_colors, colors), d_opac, None if has_cov else d_scales,
                    None if has_cov else d_rots, d_cov if has_cov else None, None)
        dev = means3D.device
        P, M = means3D.shape[0], ctx.M
        e = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)   # kernels write every element
        out = dict(means2D=e(P, 3), opacities=e(P, 1), colors=e(P, 3) if colors is not None else None,
                   means3D=e(P, 3), cov3D=e(P, 6) if cov is not None else None,
                   sh=e(P, M, 3) if sh is not None else None, scales=e(P, 3) if cov is None else None,
                   rots=e(P, 4) if cov is None else None)
        raw = dict(num_rendered=ctx.num_rendered, M=M, settings=rs, flags=0,
                   inputs=(means3D, sh, colors, scales, rots, cov), opacities=None, consts=ctx.consts, radii=radii,
                   geom=geomB, binning=binB, image=imgB)
        args = (ctx.consts[0], means3D, radii, colors, scales, rots, rs.scale_modifier, cov, ctx.consts[1], ctx.consts[2],
                rs.tanfovx, rs.tanfovy, grad_out_color, sh, rs.sh_degree, ctx.consts[3], geomB, ctx.num_rendered, binB,
                imgB, rs.debug)
        _debug_guard(rs.debug, args, "snapshot_bw.dump", "backward", lambda: backward_raw(raw, grad_out_color, out))
        return (out["means3D"], out["means2D"], out["sh"], out["colors"], out["opacities"], out["scales"], out["rots"],
                out["cov3D"], None)


def backward_raw(raw, grad_out_color, out, flags=None, grad_acc=None):
    """One call of e3dgs_rasterize_backward for a forward_raw() result.  `out` maps
    means2D/opacities/colors/means3D/cov3D/sh/scales/rots to destination tensors (or None);
    with FLAG_ACCUMULATE they are added to, otherwise fully overwritten."""
    L = _lib.lib()
    rs = raw["settings"]
    flags = raw["flags"] if flags is None else flags
    means3D, sh, colors, scales, rots, cov = raw["inputs"]
    bg, view, proj, campos = raw["consts"]
    dev = means3D.device
    P = means3D.shape[0]
    if P == 0:
        for t in out.values():
            if t is not None and not (flags & _lib.FLAG_ACCUMULATE):
                t.zero_()
        return
    H, W = int(rs.image_height), int(rs.image_width)
    g = grad_out_color
    if g.dtype != torch.float32:
        g = g.float()
    g = g.contiguous()
    if grad_acc is None:      # one record per (tile, Gaussian) instance + one sum per Gaussian; all written by the op
        grad_acc = torch.empty(int(raw["num_rendered"]) + P, _lib.ACC_STRIDE, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = L.e3dgs_rasterize_backward(
            P, int(rs.sh_degree), raw["M"], raw["num_rendered"], _lib.ptr(bg), W, H, _lib.ptr(means3D), _lib.ptr(sh),
            _lib.ptr(colors), _lib.ptr(raw.get("opacities")), _lib.ptr(scales), float(rs.scale_modifier),
            _lib.ptr(rots), _lib.ptr(cov), _lib.ptr(view), _lib.ptr(proj), _lib.ptr(campos), float(rs.tanfovx),
            float(rs.tanfovy), _lib.ptr(raw["radii"]), _lib.ptr(raw["geom"]), _lib.ptr(raw["binning"]),
            _lib.ptr(raw["image"]), _lib.ptr(g), _lib.ptr(grad_acc), _lib.ptr(out.get("means2D")),
            _lib.ptr(out.get("opacities")), _lib.ptr(out.get("colors")), _lib.ptr(out.get("means3D")),
            _lib.ptr(out.get("cov3D")), _lib.ptr(out.get("sh")), _lib.ptr(out.get("scales")), _lib.ptr(out.get("rots")),
            int(bool(rs.debug)), int(flags), _lib.current_stream())
    _lib.check(rc, "e3dgs_rasterize_backward")


# ---------------------------------------------------------------- several views of the same Gaussians in one pass
def _view_arrays(settings_list):
    """Host arrays for the multi-view entry points + the tensors that must stay alive."""
    import ctypes as C
    n = len(settings_list)
    views = [_prep(rs.viewmatrix, "viewmatrix") for rs in settings_list]
    projs = [_prep(rs.projmatrix, "projmatrix") for rs in settings_list]
    camps = [_prep(rs.campos, "campos") for rs in settings_list]
    pp = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    tx = (C.c_float * n)(*[float(rs.tanfovx) for rs in settings_list])
    ty = (C.c_float * n)(*[float(rs.tanfovy) for rs in settings_list])
    return (pp(views), pp(projs), pp(camps), tx, ty), (views, projs, camps)


def _check_same_frame(settings_list):
    r0 = settings_list[0]
    for rs in settings_list[1:]:
        if (int(rs.image_height), int(rs.image_width)) != (int(r0.image_height), int(r0.image_width)) or \
                float(rs.scale_modifier) != float(r0.scale_modifier) or int(rs.sh_degree) != int(r0.sh_degree):
            raise ValueError("the views of one multi-view call share resolution, scale_modifier and SH degree")


def forward_multi_begin(means3D, sh, opacities, scales, rotations, settings_list, flags=0, count_host=None, pool=None):
    """Enqueue-only first half of e3dgs_rasterize_forward_multi for len(settings_list) cameras (same frame size,
    background of settings_list[0]).  `count_host`: pinned int32[1] receiving the total instance count."""
    L = _lib.lib()
    _check_same_frame(settings_list)
    rs = settings_list[0]
    n = len(settings_list)
    dev = means3D.device
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a CUDA/HIP tensor (this op has no CPU path)")
    P = means3D.shape[0]
    H, W = int(rs.image_height), int(rs.image_width)
    means3D_c, sh_c, opac_c = _prep(means3D, "means3D"), _prep(sh, "shs"), _prep(opacities, "opacities")
    scales_c, rots_c = _prep(scales, "scales"), _prep(rotations, "rotations")
    bg = _prep(rs.bg, "bg")
    M = sh.shape[0] // 3 if (flags & _lib.FLAG_SH_PLANAR) else sh.shape[1]       # (sh_c is None when P == 0)
    if count_host is None:
        count_host = torch.zeros(1, dtype=torch.int32).pin_memory()
    arrays, keep = _view_arrays(settings_list)
    if flags & _lib.FLAG_COUNT_MAPPED:
        count_host[0] = -1                      # sentinel: the GPU overwrites it with the count (wait_count polls)
    if pool is not None:
        radii = pool.typed("radii", (n, P), torch.int32)
    else:
        radii = torch.empty(n, P, dtype=torch.int32, device=dev)
    geom, img = _Scratch(dev, pool, "geom"), _Scratch(dev, pool, "image")
    with torch.cuda.device(dev):
        rc = L.e3dgs_rasterize_forward_multi_begin(
            geom.cb, None, img.cb, None, n, P, int(rs.sh_degree), M, W, H, _lib.ptr(means3D_c), _lib.ptr(sh_c), None,
            _lib.ptr(opac_c), _lib.ptr(scales_c), float(rs.scale_modifier), _lib.ptr(rots_c), None, *arrays,
            _lib.ptr(radii), int(bool(rs.debug)), int(flags), count_host.data_ptr(), _lib.current_stream())
    _lib.check(rc, "e3dgs_rasterize_forward_multi_begin")
    return PendingForward(rs=rs, settings_list=list(settings_list), n=n, flags=int(flags), P=P, W=W, H=H, M=M,
                          radii=radii, geom=geom.tensor, image=img.tensor,
                          inputs=(means3D_c, sh_c, None, scales_c, rots_c, None), opacities=opac_c, bg=bg, keep=keep,
                          count_host=count_host, device=dev, pool=pool)


def wait_count(pending, timeout_s=10.0):
    """Wait until the instance count of a forward_multi_begin(..., flags | FLAG_COUNT_MAPPED) call has arrived in its
    pinned host word: the GPU stores it there directly and this polls it (a stream synchronisation costs a copy
    command plus the runtime's wake-up latency, during which the GPU has nothing queued)."""
    import ctypes as C
    import time
    if isinstance(pending, dict):            # a forward_multi_capacity() result: always mapped
        count_host, device = pending["count_host"], pending["device"]
    else:
        if not (pending.flags & _lib.FLAG_COUNT_MAPPED):
            torch.cuda.current_stream(pending.device).synchronize()
            return int(pending.count_host[0])
        count_host, device = pending.count_host, pending.device
    word = C.c_int.from_address(count_host.data_ptr())
    spins, t0 = 0, None
    while word.value == -1:
        spins += 1
        if spins & 0xFFFF == 0:
            if t0 is None:
                t0 = time.perf_counter()
            elif time.perf_counter() - t0 > timeout_s:
                torch.cuda.current_stream(device).synchronize()      # surfaces a device error, if any
                if word.value == -1:
                    raise RuntimeError("instance count never arrived")
    return int(word.value)


def prepare_multi_finish(pending):
    """Everything forward_multi_finish needs except the instance count: called BEFORE waiting for the count so that
    the host has as little as possible to do between the count's arrival and the next kernel launch (the GPU is
    idle in that window)."""
    L = _lib.lib()
    p = pending
    if p.pool is not None:
        out_color = p.pool.typed("out_color", (p.n, 3, p.H, p.W))
    else:
        out_color = torch.empty(p.n, 3, p.H, p.W, dtype=torch.float32, device=p.device)
    binning = _Scratch(p.device, p.pool, "binning")
    fixed_a = (binning.cb, None, p.n, p.P, p.W, p.H, _lib.ptr(p.bg), _lib.ptr(p.geom), _lib.ptr(p.image))
    fixed_b = (_lib.ptr(out_color), int(bool(p.rs.debug)))
    stream = _lib.current_stream()
    if p.flags & _lib.FLAG_DEFER_COLOR:
        # SH -> RGB right before compositing; `before_colour` (a Python callable set on the pending object) runs on
        # the host just before that kernel is enqueued, e.g. to make the stream wait for the SH coefficients' owner
        import ctypes as C
        means3D_c, sh_c = p.inputs[0], p.inputs[1]
        campos = (C.c_void_p * p.n)(*[t.data_ptr() for t in p.keep[2]])
        hook = getattr(p, "before_colour", None)
        notify = _lib.NOTIFY_FN((lambda _u: hook()) if hook else (lambda _u: None))
        p._colour_keep = (campos, notify)
        fn = L.e3dgs_rasterize_forward_multi_finish_colour
        tail = (int(p.rs.sh_degree), p.M, _lib.ptr(means3D_c), _lib.ptr(sh_c), campos,
                p.flags & (_lib.FLAG_SH_PLANAR | _lib.OPTION_MASK), notify, None, stream)
        p.prepared = (out_color, binning, lambda count: fn(*fixed_a, count, *fixed_b, *tail))
    else:
        fn = L.e3dgs_rasterize_forward_multi_finish
        opts = p.flags & _lib.OPTION_MASK
        p.prepared = (out_color, binning, lambda count: fn(*fixed_a, count, *fixed_b, opts, stream))
    return p.prepared


def forward_multi_finish(pending):
    """Second half; the caller has synchronised the stream since forward_multi_begin().  Returns a dict like
    forward_raw's with color (n,3,H,W) and radii (n,P)."""
    L = _lib.lib()
    p = pending
    out_color, binning, call = p.prepared if getattr(p, "prepared", None) else prepare_multi_finish(p)
    I = int(p.count_host[0])
    with torch.cuda.device(p.device):
        rc = call(I)
    _lib.check(rc, "e3dgs_rasterize_forward_multi_finish")
    return dict(color=out_color, radii=p.radii, num_rendered=I, M=p.M, settings=p.rs, settings_list=p.settings_list,
                flags=p.flags & ~(_lib.FLAG_COUNT_MAPPED | _lib.FLAG_DEFER_COLOR), inputs=p.inputs,
                opacities=p.opacities, bg=p.bg, keep=p.keep,
                geom=p.geom, binning=binning.tensor, image=p.image, pool=p.pool)


def forward_multi_capacity(means3D, sh, opacities, scales, rotations, settings_list, capacity, count_host, flags=0,
                           pool=None, before_colour=None):
    """e3dgs_rasterize_forward_multi_capacity: the whole multi-view forward enqueued in one go, binning buffers sized for
    `capacity` instances before the count is known (no host wait).  `count_host`: pinned int32[1]; it is armed with -1 here
    and receives the instance count from the GPU -- poll it with wait_count(raw) before anything persistent consumes
    the results; a count above `capacity` means nothing was emitted (repeat with a larger capacity).  The returned dict
    has num_rendered = capacity (the scratch layouts backward_multi must use) and flags | FLAG_COUNT_DEVICE."""
    import ctypes as C
    L = _lib.lib()
    _check_same_frame(settings_list)
    rs = settings_list[0]
    n = len(settings_list)
    dev = means3D.device
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a CUDA/HIP tensor (this op has no CPU path)")
    P = means3D.shape[0]
    H, W = int(rs.image_height), int(rs.image_width)
    means3D_c, sh_c, opac_c = _prep(means3D, "means3D"), _prep(sh, "shs"), _prep(opacities, "opacities")
    scales_c, rots_c = _prep(scales, "scales"), _prep(rotations, "rotations")
    bg = _prep(rs.bg, "bg")
    M = sh.shape[0] // 3 if (flags & _lib.FLAG_SH_PLANAR) else sh.shape[1]
    arrays, keep = _view_arrays(settings_list)
    flags = int(flags) | _lib.FLAG_COUNT_MAPPED
    count_host[0] = -1
    if pool is not None:
        radii = pool.typed("radii", (n, P), torch.int32)
        out_color = pool.typed("out_color", (n, 3, H, W))
    else:
        radii = torch.empty(n, P, dtype=torch.int32, device=dev)
        out_color = torch.empty(n, 3, H, W, dtype=torch.float32, device=dev)
    geom, binning, img = _Scratch(dev, pool, "geom"), _Scratch(dev, pool, "binning"), _Scratch(dev, pool, "image")
    notify = _lib.NOTIFY_FN((lambda _u: before_colour()) if before_colour else (lambda _u: None))
    with torch.cuda.device(dev):
        rc = L.e3dgs_rasterize_forward_multi_capacity(
            geom.cb, None, binning.cb, None, img.cb, None, n, P, int(rs.sh_degree), M, _lib.ptr(bg), W, H,
            _lib.ptr(means3D_c), _lib.ptr(sh_c), _lib.ptr(opac_c), _lib.ptr(scales_c), float(rs.scale_modifier),
            _lib.ptr(rots_c), *arrays, _lib.ptr(out_color), _lib.ptr(radii), int(bool(rs.debug)), flags, int(capacity),
            count_host.data_ptr(), notify, None, _lib.current_stream())
    _lib.check(rc, "e3dgs_rasterize_forward_multi_capacity")
    if P == 0:
        count_host[0] = 0          # (no kernel runs for an empty scene: nothing would ever store the count)
    return dict(color=out_color, radii=radii, num_rendered=int(capacity) if P else 0, capacity=int(capacity), M=M,
                settings=rs, settings_list=list(settings_list),
                flags=(flags & ~(_lib.FLAG_COUNT_MAPPED | _lib.FLAG_DEFER_COLOR)) | _lib.FLAG_COUNT_DEVICE,
                inputs=(means3D_c, sh_c, None, scales_c, rots_c, None), opacities=opac_c, bg=bg, keep=keep,
                geom=geom.tensor, binning=binning.tensor, image=img.tensor, pool=pool, count_host=count_host, device=dev)


def forward_multi(means3D, sh, opacities, scales, rotations, settings_list, flags=0, pool=None):
    """begin + synchronise + finish."""
    pend = forward_multi_begin(means3D, sh, opacities, scales, rotations, settings_list, flags, pool=pool)
    prepare_multi_finish(pend)
    wait_count(pend)
    return forward_multi_finish(pend)


LUV_WEIGHTS = (0.4124, 0.35758, 0.1804)     # rgb_to_LUVscale, utils/loss_utils.py:24-28 (the event contrast renders)
GRAY_WEIGHTS = (0.299, 0.587, 0.114)        # rgb_to_grayscale, utils/loss_utils.py:18-23 (the --gray losses)


def backward_multi(raw, grad_out_color, out, flags=None, grad_acc=None, stats_grad_view0=None, rank1=None):
    """e3dgs_rasterize_backward_multi for a forward_multi result.  grad_out_color is (n,3,H,W); `out` maps
    means2D (optional: (P,3) = view 0's screen-space gradient, (n,P,3) = every view's) / opacities / means3D / sh / scales /
    rots to tensors that are fully overwritten with the gradient summed over the views.  Optional `colour_views` (n,P,3): the per-view clamp-masked colour
    gradients (then `sh` may be omitted; see sh_grad_from_colour).
    stats_grad_view0 (3,H,W): e3dgs_rasterize_backward_multi_stats -- out["means2D"] is the screen-space gradient view 0
    has under THAT pixel gradient, everything else follows grad_out_color (shared-pose iterations that collect
    densification statistics).
    rank1: {view index: (w0, w1, w2)} -- e3dgs_rasterize_backward_multi_rank1: the pixel gradient of those views is
    grad_out_color[v, 0] * w (planes 1, 2 of the view are not read): a loss on a luminance of the render."""
    L = _lib.lib()
    rs = raw["settings"]
    sl = raw["settings_list"]
    flags = raw["flags"] if flags is None else flags
    means3D, sh, _, scales, rots, _ = raw["inputs"]
    dev = means3D.device
    P = means3D.shape[0]
    if P == 0:
        for t in out.values():
            if t is not None:
                t.zero_()
        return
    H, W = int(rs.image_height), int(rs.image_width)
    g = grad_out_color
    if g.dtype != torch.float32:
        g = g.float()
    g = g.contiguous()
    if tuple(g.shape) != (len(sl), 3, H, W):
        raise ValueError("grad_out_color must be (nviews,3,H,W)")
    if grad_acc is None and raw.get("pool") is not None:
        grad_acc = raw["pool"].typed("grad_acc", (int(raw["num_rendered"]) + len(sl) * P, _lib.ACC_STRIDE))
    elif grad_acc is None:
        grad_acc = torch.empty(int(raw["num_rendered"]) + len(sl) * P, _lib.ACC_STRIDE, dtype=torch.float32,
                               device=dev)
    m2 = out.get("means2D")
    if m2 is not None and m2.dim() == 3:
        # (nviews, P, 3): every view's screen-space gradient (E3DGS_FLAG_MEAN2D_VIEWS); (P, 3): view 0's only
        if tuple(m2.shape) != (len(sl), P, 3) or not m2.is_contiguous():
            raise ValueError("a per-view means2D output must be a contiguous (nviews, P, 3) tensor")
        flags = int(flags) | _lib.FLAG_MEAN2D_VIEWS
    arrays, keep = _view_arrays(sl)
    head = (len(sl), P, int(rs.sh_degree), raw["M"], raw["num_rendered"], _lib.ptr(raw["bg"]), W, H, _lib.ptr(means3D),
            _lib.ptr(sh), _lib.ptr(raw.get("opacities")), _lib.ptr(scales), float(rs.scale_modifier), _lib.ptr(rots),
            *arrays, _lib.ptr(raw["radii"]), _lib.ptr(raw["geom"]), _lib.ptr(raw["binning"]), _lib.ptr(raw["image"]))
    tail = (_lib.ptr(grad_acc), _lib.ptr(out.get("means2D")), _lib.ptr(out.get("opacities")),
            _lib.ptr(out.get("means3D")), _lib.ptr(out.get("sh")), _lib.ptr(out.get("scales")),
            _lib.ptr(out.get("rots")), _lib.ptr(out.get("colour_views")), int(bool(rs.debug)), int(flags),
            _lib.current_stream())
    g2 = stats_grad_view0
    if g2 is not None and not (g2.is_cuda and g2.dtype == torch.float32 and g2.is_contiguous()
                               and tuple(g2.shape) == (3, H, W)):
        raise ValueError("stats_grad_view0 must be a contiguous fp32 (3,H,W) GPU tensor")
    with torch.cuda.device(dev):
        if rank1:
            import ctypes as C
            wts = (C.c_float * (3 * len(sl)))()
            mask = 0
            for v, w in rank1.items():
                if not 0 <= int(v) < len(sl):
                    raise ValueError("rank1 names a view the call does not have")
                mask |= 1 << int(v)
                wts[3 * int(v)], wts[3 * int(v) + 1], wts[3 * int(v) + 2] = (float(x) for x in w)
            rc = L.e3dgs_rasterize_backward_multi_rank1(*head, _lib.ptr(g), _lib.ptr(g2), wts, mask, *tail)
        elif g2 is None:
            rc = L.e3dgs_rasterize_backward_multi(*head, _lib.ptr(g), *tail)
        else:
            rc = L.e3dgs_rasterize_backward_multi_stats(*head, _lib.ptr(g), _lib.ptr(g2), *tail)
    _lib.check(rc, "e3dgs_rasterize_backward_multi")


def sh_grad_from_colour(means3D, packed, nranks, views_per_rank, sh_degree, M, dL_dsh, scale, planar=True):
    """e3dgs_sh_grad_from_colour: the mean SH gradient over all ranks' views, rebuilt from the all-gathered per-view
    colour gradients.  `packed` is (nranks, views_per_rank*P*3 + views_per_rank*3) float32: per rank the
    `colour_views` output of backward_multi followed by its camera centres."""
    L = _lib.lib()
    P = means3D.shape[0]
    if packed.dim() != 2 or packed.shape[0] != nranks or not packed.is_contiguous():
        raise ValueError("packed must be a contiguous (nranks, block) tensor")
    with torch.cuda.device(means3D.device):
        rc = L.e3dgs_sh_grad_from_colour(P, nranks, views_per_rank, int(sh_degree), int(M), _lib.ptr(means3D),
                                         _lib.ptr(packed), packed.shape[1], float(scale), _lib.ptr(dL_dsh),
                                         _lib.FLAG_SH_PLANAR if planar else 0, _lib.current_stream())
    _lib.check(rc, "e3dgs_sh_grad_from_colour")


def sh_adam_from_colour(means3D, packed, nranks, views_per_rank, sh_degree, M, sh, exp_avg, exp_avg_sq, lr_dc, lr_rest,
                        step, scale=1.0, beta1=0.9, beta2=0.999, eps=1e-15, planar=True, mean_grad=None):
    """e3dgs_sh_adam_from_colour: rebuild of the (mean) SH gradient from per-view colour gradients fused with the Adam
    update of the SH coefficients `sh` (in place, with their moments).  `packed` as for sh_grad_from_colour.
    mean_grad (P,3): e3dgs_sh_adam_from_colour_mean -- the position gradient of a backward_multi that ran with
    FLAG_DEFER_SH_MEAN; the term through the SH view directions is added to it."""
    L = _lib.lib()
    P = means3D.shape[0]
    if packed.dim() != 2 or packed.shape[0] != nranks or not packed.is_contiguous():
        raise ValueError("packed must be a contiguous (nranks, block) tensor")
    head = (P, nranks, views_per_rank, int(sh_degree), int(M), _lib.ptr(means3D), _lib.ptr(packed), packed.shape[1],
            float(scale), _lib.ptr(sh), _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq), float(lr_dc), float(lr_rest), beta1, beta2,
            eps, int(step), _lib.FLAG_SH_PLANAR if planar else 0)
    with torch.cuda.device(means3D.device):
        if mean_grad is None:
            rc = L.e3dgs_sh_adam_from_colour(*head, _lib.current_stream())
        else:
            if not (mean_grad.is_cuda and mean_grad.dtype == torch.float32 and mean_grad.is_contiguous()
                    and tuple(mean_grad.shape) == (P, 3)):
                raise ValueError("mean_grad must be a contiguous fp32 (P,3) GPU tensor")
            rc = L.e3dgs_sh_adam_from_colour_mean(*head, _lib.ptr(mean_grad), _lib.current_stream())
    _lib.check(rc, "e3dgs_sh_adam_from_colour")


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            pos = _prep(positions, "positions")
            if pos is None or pos.shape[0] == 0:
                return torch.zeros(0, dtype=torch.bool, device=positions.device)
            present = _mark_visible(pos, _prep(rs.viewmatrix, "viewmatrix"), _prep(rs.projmatrix, "projmatrix"))
        return present.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, rs)
