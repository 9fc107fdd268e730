"""The adoption ladder: what a maintainer of the reference's `train.py` swaps in, one line at a time, and what each
swap buys (bench.py: `dropin_autograd_step.ladder`; INTEGRATION.md "Adoption ladder").

    rung 0  nothing changed: the two drop-in packages on sys.path, the reference's own render() / loss / optimizer
    rung 1  `from event_3dgs_amd.adopt import render`                  (gaussian_renderer/__init__.py:20-104)
    rung 2  + the loss block train.py:165-203 -> `adopt.event_loss(...)`
    rung 3  + `gaussians.optimizer = adopt.FusedAdam(l, lr=0.0, eps=1e-15)`  (scene/gaussian_model.py:154-163)
              and `optimizer_c = adopt.FusedAdam([c], lr=0.1)`               (train.py:71-73)
    rung 4  + the three render() calls train.py:144,159,161 -> ONE `adopt.render_views((cam, cam_now, cam_next), ...)`
              (one multi-view pass of the rasteriser, forward and backward, inside one autograd node)

Everything stays inside torch autograd and torch.optim's interfaces -- `loss.backward()`, `optimizer.step()`,
`optimizer.state[p]["exp_avg"]` (which the reference's densification code edits, scene/gaussian_model.py:258-347),
`viewspace_point_tensor.grad` -- so the rest of train.py (densification, logging, checkpoints) runs unchanged.
The fused `EventTrainer.step` (one multi-view pass, no autograd graph) remains the fast path; this module is the
measured way there.
"""
import math

import torch

from . import _lib, losses, rasterizer
from .rasterizer import GaussianRasterizationSettings

_RAW_ATTRS = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")


def _reference_activations(pc):
    """True when `pc` carries the reference's activation functions (scene/gaussian_model.py:33-41) -- or none at all
    (renderer.GaussianView): the in-kernel activations of E3DGS_FLAG_PREACT are exactly those."""
    ok = True
    for name, fn in (("scaling_activation", torch.exp), ("opacity_activation", torch.sigmoid),
                     ("rotation_activation", torch.nn.functional.normalize)):
        have = getattr(pc, name, None)
        ok = ok and (have is None or have is fn)
    return ok


def _features(pc):
    """(P, 16, 3) SH coefficients of the model: torch.cat((_features_dc, _features_rest), 1) as get_features does
    (scene/gaussian_model.py:105-108), made ONCE per parameter version instead of once per render() -- the three renders of
    an iteration see the same coefficients (192 MB per cat at 1 M Gaussians).  The cached tensor is detached; the
    gradient goes back to the two leaves through the autograd node of render()."""
    import weakref
    dc, rest = pc._features_dc, pc._features_rest
    key = (dc._version, rest._version, dc.data_ptr(), rest.data_ptr(), tuple(dc.shape), tuple(rest.shape))
    hit = getattr(pc, "_e3dgs_feature_cache", None)
    # (weak references: after a densification step the entry must not keep the replaced Parameters -- and their 192 MB at
    # 1 M Gaussians -- alive until the next render.  Writers that go through raw pointers must bump the version counters:
    # FusedAdam and losses.adam_step_ / adam_step_segments_ do; anything else calls adopt.invalidate(pc).)
    if hit is not None and hit[0] == key and hit[1]() is dc and hit[2]() is rest:
        return hit[3]
    with torch.no_grad():
        cat = torch.cat((dc, rest), dim=1).contiguous()
    try:
        pc._e3dgs_feature_cache = (key, weakref.ref(dc), weakref.ref(rest), cat)
    except AttributeError:
        pass
    return cat


def invalidate(pc):
    """Drop the cached (P, 16, 3) coefficient tensor of `pc` (see _features): for callers that update `_features_dc` /
    `_features_rest` through a path that bumps no autograd version counter (a raw-pointer kernel of their own, DLPack)."""
    try:
        pc._e3dgs_feature_cache = None
    except AttributeError:
        pass


class _SplitFeatures(torch.autograd.Function):
    """Identity on the cached (P, 16, 3) coefficient tensor whose backward hands the gradient's [:, :1] / [:, 1:] slices to
    the two leaves -- the backward of the torch.cat that was not recorded."""

    @staticmethod
    def forward(ctx, dc, rest, cat):
        ctx.n_dc = dc.shape[1]
        return cat.view_as(cat)

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.n_dc], g[:, ctx.n_dc:], None


import os as _os

DEFER_RENDERS = _os.environ.get("E3DGS_ADOPT_DEFER", "1") != "0"
_META_GETTERS = None


def _meta_getters():
    """Tensor attributes a lazy result answers from its own metadata (shape, dtype, device ...) without rendering."""
    global _META_GETTERS
    if _META_GETTERS is None:
        T = torch.Tensor
        _META_GETTERS = {T.shape.__get__, T.dtype.__get__, T.device.__get__, T.is_cuda.__get__, T.ndim.__get__, T.size,
                         T.dim, T.numel, T.element_size, T.is_floating_point, T.is_contiguous, T.stride, T.__len__,
                         T.layout.__get__, T.is_sparse.__get__, T.is_quantized.__get__, T.is_meta.__get__}
    return _META_GETTERS


class _Lazy(torch.Tensor):
    """One output of a deferred render(): a tensor whose VALUE exists from its first use on.  Any torch operation on it
    (operators, methods, torch.* functions, indexing with it) is run on the real tensor -- which carries the autograd
    history of the rasteriser node -- after the pending renders of its batch have gone through the rasteriser in one
    pass.  Shape / dtype / device are answered without rendering."""

    @staticmethod
    def __new__(cls, batch, getter, shape, dtype, device):
        t = torch.Tensor._make_wrapper_subclass(cls, tuple(shape), dtype=dtype, device=device, requires_grad=False)
        t._e3_batch, t._e3_get = batch, getter
        return t

    def _value(self):
        self._e3_batch.flush()
        return self._e3_get()

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _meta_getters():
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        from torch.utils._pytree import tree_map
        real = lambda x: x._value() if isinstance(x, _Lazy) else x
        return func(*tree_map(real, args), **tree_map(real, kwargs))

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # (reached only when something bypasses __torch_function__, e.g. a C++ extension that takes the tensor as it is)
        from torch.utils._pytree import tree_map
        real = lambda x: x._value() if isinstance(x, _Lazy) else x
        return func(*tree_map(real, args), **tree_map(real, kwargs or {}))


def materialize(x):
    """The real tensor behind a deferred result (anything else is returned as it is).  adopt's own loss functions call it;
    a caller who hands a deferred image to a torch.autograd.Function of their OWN must too -- Function.apply takes its
    arguments without dispatching, so the function would see a tensor with no history and the gradient would be lost."""
    return x._value() if isinstance(x, _Lazy) else x


class _PendingRenders:
    """The renders issued on one parameter version of one model that nothing has used yet."""
    MAX_VIEWS = 4

    def __init__(self, pc, key):
        self.pc, self.key = pc, key
        self.items = []                 # (settings, screenspace_points)
        self.done = False
        self.images = self.radii = None

    @staticmethod
    def key_of(pc, rs):
        ts = (pc._xyz, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling, pc._rotation)
        return (tuple((id(t), t._version, t.data_ptr()) for t in ts), id(rs.bg), rs.bg._version, float(rs.scale_modifier),
                int(rs.sh_degree), int(rs.image_height), int(rs.image_width))

    def versions_now(self):
        pc = self.pc
        ts = (pc._xyz, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling, pc._rotation)
        return tuple((id(t), t._version, t.data_ptr()) for t in ts)

    def flush(self):
        if self.done:
            return
        pc = self.pc
        if self.versions_now() != self.key[0]:
            raise RuntimeError("the model's parameters were replaced or modified in place between render() and the first use "
                               "of its result: a deferred render cannot reproduce the old values (use the image before the "
                               "optimizer step, or set E3DGS_ADOPT_DEFER=0 / render(..., defer=False))")
        if getattr(pc, "_e3dgs_pending", None) is self:
            pc._e3dgs_pending = None
        self.done = True
        with torch.enable_grad():           # (first use may sit inside a no_grad block -- a logging line: the graph is still built)
            shs = _features(pc)
            if pc._features_dc.requires_grad or pc._features_rest.requires_grad:
                shs = _SplitFeatures.apply(pc._features_dc, pc._features_rest, shs)
            if len(self.items) == 1:
                rs, ssp = self.items[0]
                image, radii = rasterizer.rasterize_gaussians(pc._xyz, ssp, shs, None, pc._opacity, pc._scaling, pc._rotation,
                                                              None, rs, flags=_lib.FLAG_PREACT)
                self.images, self.radii = [image], [radii]
            else:
                settings = tuple(rs for rs, _ in self.items)
                radii, *images = _RasterizeViews.apply(pc._xyz, shs, pc._opacity, pc._scaling, pc._rotation, settings,
                                                       *[ssp for _, ssp in self.items])
                self.images, self.radii = list(images), [radii[k] for k in range(len(self.items))]
        self.items = [(rs, None) for rs, _ in self.items]


def _deferred_render(pc, rs, screenspace_points):
    """render() without running the rasteriser yet.  The renders pending on the SAME parameter version of `pc` (same
    background tensor, scale modifier, SH degree and frame size; at most four) are rendered together by the first use of
    any of their results: one multi-view pass, forward and backward, instead of one pass each.

    What differs from three immediate renders, and nothing else (every render's `viewspace_points.grad` is filled with its
    own view's screen-space gradient, E3DGS_FLAG_MEAN2D_VIEWS):
      * results are `_Lazy` tensors until first use -- every torch operation sees the real tensor; a
        torch.autograd.Function of the caller's own must be given adopt.materialize(image);
      * an in-place parameter update between render() and the first use raises instead of rendering stale or new values."""
    key = _PendingRenders.key_of(pc, rs)
    batch = getattr(pc, "_e3dgs_pending", None)
    if batch is None or batch.done or batch.key != key or len(batch.items) >= _PendingRenders.MAX_VIEWS:
        batch = _PendingRenders(pc, key)
        try:
            pc._e3dgs_pending = batch
        except AttributeError:                  # (a model object that takes no attributes: render now)
            batch.items.append((rs, screenspace_points))
            batch.flush()
            return {"render": batch.images[0], "viewspace_points": screenspace_points,
                    "visibility_filter": batch.radii[0] > 0, "radii": batch.radii[0]}
    k = len(batch.items)
    batch.items.append((rs, screenspace_points))
    dev, P = pc._xyz.device, pc._xyz.shape[0]
    H, W = int(rs.image_height), int(rs.image_width)
    return {"render": _Lazy(batch, lambda: batch.images[k], (3, H, W), torch.float32, dev),
            "viewspace_points": screenspace_points,
            "visibility_filter": _Lazy(batch, lambda: batch.radii[k] > 0, (P,), torch.bool, dev),
            "radii": _Lazy(batch, lambda: batch.radii[k], (P,), torch.int32, dev)}


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, defer=None):
    """Rung 1: the reference's render() (gaussian_renderer/__init__.py:20-104) -- same arguments, same result dict
    {"render", "viewspace_points", "visibility_filter", "radii"} -- with everything between the model's raw parameters
    and the image inside the rasteriser: SH evaluation (the reference forces the torch branch at :71: ~40 elementwise
    launches per render), exp / sigmoid / normalize (E3DGS_FLAG_PREACT) and their chain rule, one autograd node in C++.
    Needs the reference model's raw tensors (`pc._xyz`, `_features_dc`, `_features_rest`, `_scaling`, `_rotation`,
    `_opacity`) and its activations; anything else (override_color, pipe.compute_cov3D_python, pipe.debug, a model with
    other activations, no compiled extension) goes through renderer.render, i.e. the reference's own sequence.

    DEFERRED (defer=True; default: adopt.DEFER_RENDERS, environment E3DGS_ADOPT_DEFER, on): with gradients enabled the call
    returns at once and the rasteriser runs when an image, `radii` or `visibility_filter` of the result is first USED --
    train.py:144,159,161 issues the three renders of an event iteration before any loss operation reads them, so they go
    through the rasteriser as ONE multi-view pass (what render_views does explicitly), forward and backward.  See
    _deferred_render for what that changes (results are lazy until first use; a parameter update in between raises)."""
    from . import renderer
    fast = (override_color is None and not getattr(pipe, "compute_cov3D_python", False)
            and not getattr(pipe, "debug", False) and all(hasattr(pc, a) for a in _RAW_ATTRS)
            and _reference_activations(pc) and rasterizer.cpp_autograd_ext() is not None and pc._xyz.is_cuda
            and int(pc.max_sh_degree) <= 4)
    if not fast:
        return renderer.render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color)
    xyz = pc._xyz
    screenspace_points = torch.zeros_like(xyz, requires_grad=True)           # :28-32 (a leaf: .grad is filled)
    rs = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=False)
    if defer is None:
        defer = DEFER_RENDERS
    if defer and torch.is_grad_enabled() and int(pc.max_sh_degree) <= 3:
        return _deferred_render(pc, rs, screenspace_points)
    shs = _features(pc)
    if torch.is_grad_enabled() and (pc._features_dc.requires_grad or pc._features_rest.requires_grad):
        shs = _SplitFeatures.apply(pc._features_dc, pc._features_rest, shs)
    image, radii = rasterizer.rasterize_gaussians(xyz, screenspace_points, shs, None, pc._opacity, pc._scaling,
                                                  pc._rotation, None, rs, flags=_lib.FLAG_PREACT)
    return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}


_COUNT_WORDS = {}


def _count_word(device):
    """One pinned, device-mapped int32 per device for the instance count of render_views() (allocating pinned memory per
    call costs a hipHostMalloc -- milliseconds, and it synchronises the device); the forward waits for the count before it
    returns, so one word serves every call on that device."""
    key = (device.type, device.index)
    w = _COUNT_WORDS.get(key)
    if w is None:
        w = _COUNT_WORDS[key] = torch.zeros(1, dtype=torch.int32).pin_memory()
    return w


class _RasterizeViews(torch.autograd.Function):
    """Several cameras of the same model as ONE multi-view pass of the rasteriser inside one autograd node
    (e3dgs_rasterize_forward_multi / _backward_multi; raw parameters, E3DGS_FLAG_PREACT)."""

    @staticmethod
    def forward(ctx, xyz, shs, opacity, scaling, rotation, settings, *means2D):
        # means2D: the viewspace_points leaves, one per view (each receives its view's screen-space gradient), or just the
        # first view's
        pend = rasterizer.forward_multi_begin(xyz.detach(), shs.detach(), opacity.detach(), scaling.detach(),
                                              rotation.detach(), list(settings),
                                              flags=_lib.FLAG_PREACT | _lib.FLAG_COUNT_MAPPED, count_host=_count_word(xyz.device))
        rasterizer.prepare_multi_finish(pend)          # (host work done while the GPU still computes the count)
        rasterizer.wait_count(pend)
        raw = rasterizer.forward_multi_finish(pend)
        ctx.raw = raw
        ctx.n = len(settings)
        ctx.n_m2 = len(means2D)
        # (the kernels read the inputs again in backward: saved through autograd, so that an in-place update between forward
        # and backward is reported the way autograd reports it for any saved tensor, as the C++ node of single renders does)
        ctx.save_for_backward(xyz, shs, opacity, scaling, rotation)
        radii = raw["radii"]
        ctx.mark_non_differentiable(radii)
        return (radii,) + tuple(raw["color"][k] for k in range(ctx.n))

    @staticmethod
    def backward(ctx, _g_radii, *g_imgs):
        raw, n = ctx.raw, ctx.n
        if raw is None:
            raise RuntimeError("Trying to backward through render_views() a second time: the rasteriser's scratch buffers "
                               "were released by the first backward (render again; retain_graph=True is not supported here)")
        _ = ctx.saved_tensors                    # (autograd's version check of the five inputs)
        ctx.raw = None
        xyz, shs, _, scaling, rotation, _ = raw["inputs"]
        H, W = raw["color"].shape[2], raw["color"].shape[3]
        g = torch.stack([gi if gi is not None else torch.zeros(3, H, W, device=xyz.device) for gi in g_imgs]).float()
        e = lambda t: torch.empty_like(t)
        per_view = ctx.n_m2 == n and n > 1
        out = dict(means3D=e(xyz), sh=e(shs), opacities=e(raw["opacities"]), scales=e(scaling), rots=e(rotation),
                   means2D=torch.empty((n,) + tuple(xyz.shape), dtype=xyz.dtype, device=xyz.device) if per_view
                   else torch.empty_like(xyz))
        rasterizer.backward_multi(raw, g, out)
        m2 = tuple(out["means2D"][k] for k in range(n)) if per_view else \
            ((out["means2D"],) + (None,) * (ctx.n_m2 - 1) if ctx.n_m2 else ())
        return (out["means3D"], out["sh"], out["opacities"], out["scales"], out["rots"], None) + m2


def render_views(viewpoint_cameras, pc, pipe, bg_color, scaling_modifier=1.0):
    """Rung 4: the three render() calls of an event iteration (train.py:144,159,161) as ONE call -- a list of the dicts
    render() returns, one per camera -- that goes through the rasteriser as one multi-view pass, forward and backward:

        render_pkg, render_pkg_now, render_pkg_next = adopt.render_views(
            (viewpoint_cam, viewpoint_cam_now, viewpoint_cam_next), gaussians, pipe, bg)

    Every kernel of the pipeline runs once over all cameras and the per-Gaussian backward sums the views in registers
    (what EventTrainer.step runs, here behind autograd).  The cameras must share one frame size; every dict carries its own
    `viewspace_points` leaf, filled with its view's screen-space gradient (E3DGS_FLAG_MEAN2D_VIEWS).  Falls back to one
    render() per camera when the fast path of render() does not apply or the frame sizes differ."""
    cams = list(viewpoint_cameras)
    sizes = {(int(c.image_height), int(c.image_width)) for c in cams}
    fast = (len(cams) >= 1 and len(cams) <= 4 and len(sizes) == 1 and not getattr(pipe, "compute_cov3D_python", False)
            and not getattr(pipe, "debug", False) and all(hasattr(pc, a) for a in _RAW_ATTRS)
            and _reference_activations(pc) and pc._xyz.is_cuda and int(pc.max_sh_degree) <= 3)
    if not fast:
        return [render(c, pc, pipe, bg_color, scaling_modifier) for c in cams]
    xyz = pc._xyz
    screenspace_points = torch.zeros_like(xyz, requires_grad=True)
    settings = tuple(GaussianRasterizationSettings(
        image_height=int(c.image_height), image_width=int(c.image_width), tanfovx=math.tan(c.FoVx * 0.5),
        tanfovy=math.tan(c.FoVy * 0.5), bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=c.world_view_transform,
        projmatrix=c.full_proj_transform, sh_degree=pc.active_sh_degree, campos=c.camera_center, prefiltered=False,
        debug=False) for c in cams)
    shs = _features(pc)
    if torch.is_grad_enabled() and (pc._features_dc.requires_grad or pc._features_rest.requires_grad):
        shs = _SplitFeatures.apply(pc._features_dc, pc._features_rest, shs)
    leaves = [screenspace_points] + [torch.zeros_like(xyz, requires_grad=True) for _ in cams[1:]]
    radii, *images = _RasterizeViews.apply(xyz, shs, pc._opacity, pc._scaling, pc._rotation, settings, *leaves)
    return [{"render": img, "viewspace_points": leaves[k], "visibility_filter": radii[k] > 0, "radii": radii[k]}
            for k, img in enumerate(images)]


def event_loss(image, image_now, image_next, c, gt_image_intensity, image_now_gt, image_next_gt, gt_blur_image=None,
               gt_c=0.17):
    """Rung 2: the loss block of train.py:165-203 as one autograd node (losses.event_iteration_loss: the fused event-loss
    kernels; value and gradients w.r.t. the three renders and the threshold c)."""
    return losses.event_iteration_loss(materialize(image), materialize(image_now), materialize(image_next), c,
                                       gt_image_intensity, image_now_gt, image_next_gt, gt_blur_image, gt_c)


def gray_loss(image, gt_image, lambda_dssim=0.2):
    """The `--gray` loss block train.py:213-223 ((1 - lambda) l1_loss_gray + lambda (1 - ssim_gray), utils/loss_utils.py:40-48,
    368-385) with the fused SSIM kernel behind autograd."""
    return losses.gray_iteration_loss(materialize(image), gt_image, lambda_dssim)


def rgb_loss(image, gt_image, lambda_dssim=0.2):
    """The RGB loss block train.py:292-296 ((1 - lambda) l1_loss + lambda (1 - ssim), utils/loss_utils.py:270-271,388-396)."""
    return losses.rgb_iteration_loss(materialize(image), gt_image, lambda_dssim)


class FusedAdam(torch.optim.Optimizer):
    """Rung 3: torch.optim.Adam's interface and state layout over the fused Adam kernel (e3dgs_adam_step).

    `FusedAdam(params_or_groups, lr=..., betas=(0.9, 0.999), eps=...)` accepts what the reference passes to
    torch.optim.Adam (scene/gaussian_model.py:154-163: six groups with "name" and "lr", eps 1e-15; train.py:73: `[c]`,
    lr 0.1).  Per parameter the state is {"step", "exp_avg", "exp_avg_sq"} -- the keys the reference's
    replace_tensor_to_optimizer / _prune_optimizer / cat_tensors_to_optimizer read and rewrite (:258-347) -- with `step`
    counted per parameter and parameters without a gradient skipped, as torch does.  One launch per parameter tensor
    instead of torch's foreach sequence (~12 launches per group); arithmetic as torch.optim.Adam (no amsgrad, no weight
    decay, not maximize)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if g.is_sparse:
                    raise RuntimeError("FusedAdam does not support sparse gradients")
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError("FusedAdam needs contiguous fp32 GPU parameters (no CPU path)")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                # (state that the reference's densification rebuilt: tensors of the new size, possibly non-contiguous cats)
                for k in ("exp_avg", "exp_avg_sq"):
                    if not st[k].is_contiguous():
                        st[k] = st[k].contiguous()
                step = int(st["step"]) + 1
                st["step"] = step if not torch.is_tensor(st["step"]) else st["step"] + 1
                if not g.is_contiguous():
                    g = g.contiguous()
                if g.dtype != torch.float32:
                    g = g.float()
                losses.adam_step_(p.view(-1), g.view(-1), st["exp_avg"].view(-1), st["exp_avg_sq"].view(-1),
                                  float(group["lr"]), step, beta1=b1, beta2=b2, eps=float(group["eps"]))
                # (losses.adam_step_ bumps the version counters of the three tensors -- the views share them with their
                # bases --, as an in-place torch op would: render()'s per-version cache of the concatenated SH coefficients,
                # the deferred renders' staleness check and autograd's saved-tensor check rely on it)
        return loss


class LadderLoop:
    """The reference's event iteration (train.py:110,144-212,330-332) at one rung of the ladder, on a model held the way the
    reference holds it (six nn.Parameters behind GaussianModel-style getters, an optimizer with six named groups, a second
    optimizer for the threshold c).  bench.py times it per rung; tests/test_hip_adopt.py pins every rung to
    EventTrainer.step.  rung 0 = the reference's own sequence (render() with its forced torch-SH branch, torch activations,
    the loss formulas of utils/loss_utils.py in torch, torch.optim.Adam)."""

    LR = dict(xyz=1.6e-4, features_dc=2.5e-3, features_rest=2.5e-3 / 20.0, opacity=0.05, scaling=5e-3, rotation=1e-3)

    def __init__(self, rung, params, device, c_init=0.17, python_sh=True):
        from .renderer import GaussianView, PipelineParams
        from .train_step import ExponentialLR
        if rung not in (0, 1, 2, 3, 4):
            raise ValueError("rung must be 0..4")
        self.rung, self.python_sh = rung, python_sh
        self.P = {k: torch.nn.Parameter(v.detach().clone().to(device)) for k, v in params.items()}
        self.pc = GaussianView(self.P, active_sh_degree=3, max_sh_degree=3)
        self.pipe = PipelineParams()
        groups = [{"params": [self.P[k]], "lr": self.LR[k], "name": k} for k in
                  ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")]
        Opt = FusedAdam if rung >= 3 else torch.optim.Adam
        self.optimizer = Opt(groups, lr=0.0, eps=1e-15)                       # scene/gaussian_model.py:154-163
        self.c = torch.nn.Parameter(torch.tensor([c_init], device=device))
        self.optimizer_c = Opt([self.c], lr=0.1)                             # train.py:71-73
        self.xyz_lr = ExponentialLR(1.6e-4, 1.6e-6, delay_mult=0.01, max_steps=30000)
        self.iteration = 0

    def _render(self, cam, bg):
        from . import renderer
        if self.rung >= 1:
            return render(cam, self.pc, self.pipe, bg)
        return renderer.render(cam, self.pc, self.pipe, bg, force_python_sh=self.python_sh)

    def loss(self, imgs, gts, gt_blur=None):
        if self.rung >= 2:
            return event_loss(imgs[0], imgs[1], imgs[2], self.c, gts[0], gts[1], gts[2], gt_blur)
        lum = lambda im: (0.4124 * im[0] + 0.35758 * im[1] + 0.1804 * im[2]).unsqueeze(0)      # utils/loss_utils.py:24-28
        ev = lambda a, b, cc: (torch.log(lum(b) + 1e-8) - torch.log(lum(a) + 1e-8)) / cc       # :234-249
        img_diff, gt = ev(imgs[1], imgs[2], self.c), ev(gts[1], gts[2], 0.17)
        loss1, loss2 = torch.abs(img_diff - gt).mean(), torch.abs(imgs[0] - gts[0]).mean()          # train.py:165-203
        mask = (gt != 0).to(imgs[0].dtype)
        loss = (0.9 * (loss1 * mask).sum() + 0.1 * (loss2 * (1 - mask)).sum()) / (mask.sum() + (1 - mask).sum())
        if gt_blur is not None:
            loss = 0.5 * loss + 0.5 * torch.abs(imgs[0] - gt_blur).mean()
        return loss

    def step(self, cams, gts, bg, gt_blur=None):
        self.iteration += 1
        for g in self.optimizer.param_groups:                                # update_learning_rate, train.py:110
            if g["name"] == "xyz":
                g["lr"] = self.xyz_lr(self.iteration)
        if self.rung >= 4:
            pkgs = render_views(cams, self.pc, self.pipe, bg)
        else:
            pkgs = [self._render(cam, bg) for cam in cams]                   # train.py:144,159,161
        self.viewspace_points = pkgs[0]["viewspace_points"]
        loss = self.loss([p["render"] for p in pkgs], gts, gt_blur)
        loss.backward()                                                      # :211
        self.optimizer_c.step(); self.optimizer_c.zero_grad(set_to_none=True)
        self.optimizer.step(); self.optimizer.zero_grad(set_to_none=True)    # :330-332
        return loss.detach()
