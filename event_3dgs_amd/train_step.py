"""The reference's `--event` training iteration (train.py:97-332) on the HIP hot path.

One step = THREE rasteriser forward+backward passes (intensity view train.py:144, event views
:159,:161 -- SURVEY 0.3), the event + intensity loss (:165-203), backward (:211), Adam on the
59 floats/Gaussian (:330-332; groups scene/gaussian_model.py:154-163) and on the threshold c
(:71-73,:212).  Densification (:317-327) is outside the steady-state step.

MI355X-first differences from the reference loop (results equal within fp32 tolerance):
  * SH -> RGB runs inside the rasteriser's preprocess kernel (shs path) instead of ~40 torch
    elementwise launches per render (gaussian_renderer/__init__.py:74-81);
  * parameters, gradients and Adam moments live in ONE flat fp32 buffer each, so the view-parallel
    gradient exchange is a single RCCL all-reduce and the zero-fill is a single memset;
  * f_dc / f_rest are stored interleaved as the (P,16,3) tensor the rasteriser consumes (no
    torch.cat per render, gaussian_model.py:105-108) and keep their separate learning rates.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

from . import losses
from .rasterizer import GaussianRasterizationSettings, rasterize_gaussians


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """utils/general_utils.py:29-62"""
    def helper(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        if lr_delay_steps > 0:
            delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        else:
            delay_rate = 1.0
        t = np.clip(step / max_steps, 0, 1)
        log_lerp = np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)
        return delay_rate * log_lerp
    return helper


# (name, floats per Gaussian) in flat-buffer order
SEGMENTS = (("xyz", 3), ("features", 48), ("opacity", 1), ("scaling", 3), ("rotation", 4))
FLOATS_PER_GAUSSIAN = sum(n for _, n in SEGMENTS)   # 59


class EventTrainer:
    """Holds the Gaussian parameters (pre-activation, scene/gaussian_model.py:44-59) and runs steps."""

    def __init__(self, params, device, spatial_lr_scale=1.0, position_lr_init=1.6e-4, position_lr_final=1.6e-6,
                 position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=2.5e-3, opacity_lr=0.05,
                 scaling_lr=5e-3, rotation_lr=1e-3, c_init=0.17, c_lr=0.1, active_sh_degree=3, process_group=None):
        self.device = torch.device(device)
        self.N = params["xyz"].shape[0]
        N = self.N
        self.flat = torch.empty(N * FLOATS_PER_GAUSSIAN, dtype=torch.float32, device=self.device)
        self.flat_grad = torch.zeros_like(self.flat)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.views, self.seg = {}, {}
        off = 0
        shapes = {"xyz": (N, 3), "features": (N, 16, 3), "opacity": (N, 1), "scaling": (N, 3), "rotation": (N, 4)}
        feats = torch.cat((params["features_dc"], params["features_rest"]), dim=1)
        src = {"xyz": params["xyz"], "features": feats, "opacity": params["opacity"], "scaling": params["scaling"],
               "rotation": params["rotation"]}
        for name, per in SEGMENTS:
            n = N * per
            self.seg[name] = (off, n)
            p = self.flat[off:off + n].view(shapes[name])
            p.copy_(src[name].to(self.device))
            p.requires_grad_(True)
            p.grad = self.flat_grad[off:off + n].view(shapes[name])
            self.views[name] = p
            off += n
        self.c = torch.full((1,), c_init, dtype=torch.float32, device=self.device, requires_grad=True)
        self.c_opt = torch.optim.Adam([self.c], lr=c_lr)
        self.xyz_lr = get_expon_lr_func(position_lr_init * spatial_lr_scale, position_lr_final * spatial_lr_scale,
                                        lr_delay_mult=position_lr_delay_mult, max_steps=position_lr_max_steps)
        self.lrs = dict(features=feature_lr, features_rest=feature_lr / 20.0, opacity=opacity_lr, scaling=scaling_lr,
                        rotation=rotation_lr)
        self.active_sh_degree = active_sh_degree
        self.iteration = 0
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.last_stats = None

    # ---- gaussian_renderer.render() on the fused path (gaussian_renderer/__init__.py:20-104)
    def render(self, cam, bg, scaling_modifier=1.0):
        v = self.views
        rs = GaussianRasterizationSettings(
            image_height=int(cam.image_height), image_width=int(cam.image_width),
            tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg,
            scale_modifier=scaling_modifier, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
            sh_degree=self.active_sh_degree, campos=cam.camera_center, prefiltered=False, debug=False)
        means2D = torch.zeros_like(v["xyz"], requires_grad=True)
        img, radii = rasterize_gaussians(v["xyz"], means2D, v["features"], None, self._opac, self._scales, self._rots,
                                         None, rs)
        return {"render": img, "viewspace_points": means2D, "visibility_filter": radii > 0, "radii": radii}

    def _activations(self):
        v = self.views
        self._scales = torch.exp(v["scaling"])                               # gaussian_model.py:97
        self._rots = torch.nn.functional.normalize(v["rotation"])            # :101
        self._opac = torch.sigmoid(v["opacity"])                             # :117

    def step(self, cam_int, cam_now, cam_next, gt_int, gt_now, gt_next, bg, gt_blur=None, sync_grads=True):
        """One event iteration.  Returns the (device) loss tensor; no host synchronisation besides the
        rasteriser's own instance-count read-back."""
        self.iteration += 1
        it = self.iteration
        self.flat_grad.zero_()
        self.c.grad = None
        self._activations()
        r0 = self.render(cam_int, bg)
        r1 = self.render(cam_now, bg)
        r2 = self.render(cam_next, bg)
        loss = losses.event_iteration_loss(r0["render"], r1["render"], r2["render"], self.c, gt_int, gt_now, gt_next,
                                           gt_blur)
        loss.backward()
        if self.world > 1 and sync_grads:
            # view-parallel data parallelism: one all-reduce of the 59 floats/Gaussian (+ c)
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.pg)
            dist.all_reduce(self.c.grad, op=dist.ReduceOp.SUM, group=self.pg)
            self.flat_grad.div_(self.world)
            self.c.grad.div_(self.world)
        self.c_opt.step()                                                   # train.py:212
        self._adam(it)
        self.last_render = r0
        return loss

    def _adam(self, it):
        for name, _ in SEGMENTS:
            off, n = self.seg[name]
            sl = slice(off, off + n)
            if name == "xyz":
                lr, kw = self.xyz_lr(it), {}
            elif name == "features":
                lr, kw = self.lrs["features"], dict(lr_b=self.lrs["features_rest"], period=48, split=3)
            else:
                lr, kw = self.lrs[name], {}
            losses.adam_step_(self.flat[sl], self.flat_grad[sl], self.exp_avg[sl], self.exp_avg_sq[sl], lr, it, **kw)
