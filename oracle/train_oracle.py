"""The reference's `--event` training iteration on the CPU oracle (TEST INFRASTRUCTURE ONLY).

What "oracle-trained" means in the PSNR criterion: the loop body of train.py:97-332 -- three renders
(:144,159,161), event + intensity loss (:165-203), loss.backward(), optimizer_c.step() (:210-212),
gaussians.optimizer.step() (:330-332) -- with every piece the reference itself would run in PyTorch done by PyTorch on
the CPU (getters scene/gaussian_model.py:95-118, loss utils/loss_utils.py via torch_oracle, torch.optim.Adam with the
groups / learning rates of gaussian_model.py:149-167 and the xyz schedule of utils/general_utils.py:29-62) and the
rasteriser -- the one piece whose source is absent -- done by the C oracle (gs_oracle.c forward + analytic backward)
behind a torch.autograd.Function.  Nothing here imports the product packages.
"""
import math

import numpy as np
import torch

from . import c_oracle, torch_oracle


class _OracleRasterize(torch.autograd.Function):
    """diff_gaussian_rasterization's operator on the C oracle: (means3D, shs, opacities, scales, rotations) -> image."""

    @staticmethod
    def forward(ctx, means3D, shs, opacities, scales, rotations, cam, bg, sh_degree):
        f = c_oracle.Forward(means3D=means3D.detach().numpy(), opacities=opacities.detach().numpy(),
                             viewmatrix=cam["view"], projmatrix=cam["proj"], campos=cam["campos"], bg=bg,
                             width=cam["W"], height=cam["H"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                             shs=shs.detach().numpy(), sh_degree=sh_degree, scales=scales.detach().numpy(),
                             rotations=rotations.detach().numpy())
        ctx.f = f
        return torch.from_numpy(f.out_color.copy())

    @staticmethod
    def backward(ctx, grad):
        g = ctx.f.backward(grad.contiguous().numpy())
        ctx.f.close()
        t = lambda a, shape: torch.from_numpy(np.ascontiguousarray(a)).reshape(shape)
        P = g["means3D"].shape[0]
        return (t(g["means3D"], (P, 3)), t(g["shs"], (P, -1, 3)), t(g["opacities"], (P, 1)), t(g["scales"], (P, 3)),
                t(g["rotations"], (P, 4)), None, None, None)


class _OracleRasterizeViews(torch.autograd.Function):
    """The three renders of an iteration as one node: the C oracle is single-threaded and ctypes releases the GIL, so the
    views run on three host threads, forward and backward (same arithmetic per view; the gradients are summed in view
    order, as autograd would add them)."""

    @staticmethod
    def forward(ctx, means3D, shs, opacities, scales, rotations, cams, bg, sh_degree):
        from concurrent.futures import ThreadPoolExecutor
        arrs = dict(means3D=means3D.detach().numpy(), opacities=opacities.detach().numpy(), shs=shs.detach().numpy(),
                    scales=scales.detach().numpy(), rotations=rotations.detach().numpy())

        def one(cam):
            return c_oracle.Forward(viewmatrix=cam["view"], projmatrix=cam["proj"], campos=cam["campos"], bg=bg,
                                    width=cam["W"], height=cam["H"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                                    sh_degree=sh_degree, **arrs)
        with ThreadPoolExecutor(max_workers=len(cams)) as ex:
            ctx.fs = list(ex.map(one, cams))
        return tuple(torch.from_numpy(f.out_color.copy()) for f in ctx.fs)

    @staticmethod
    def backward(ctx, *grads):
        from concurrent.futures import ThreadPoolExecutor

        def one(args):
            f, g = args
            out = f.backward(g.contiguous().numpy())
            f.close()
            return out
        with ThreadPoolExecutor(max_workers=len(ctx.fs)) as ex:
            gs = list(ex.map(one, zip(ctx.fs, grads)))
        t = lambda a, shape: torch.from_numpy(np.ascontiguousarray(a)).reshape(shape)
        P = gs[0]["means3D"].shape[0]
        acc = None
        for g in gs:
            cur = [t(g["means3D"], (P, 3)), t(g["shs"], (P, -1, 3)), t(g["opacities"], (P, 1)), t(g["scales"], (P, 3)),
                   t(g["rotations"], (P, 4))]
            acc = cur if acc is None else [a + b for a, b in zip(acc, cur)]
        return tuple(acc) + (None, None, None)


def camera_dict(cam):
    """Host copy of what GaussianRasterizationSettings carries for one camera (gaussian_renderer/__init__.py:38-51)."""
    return dict(view=cam.world_view_transform.contiguous().cpu().numpy(), proj=cam.full_proj_transform.cpu().numpy(),
                campos=cam.camera_center.contiguous().cpu().numpy(), W=int(cam.image_width), H=int(cam.image_height),
                tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5))


def expon_lr(step, lr_init, lr_final, lr_delay_mult, max_steps):
    """utils/general_utils.py:29-62 with lr_delay_steps = 0 (the value training_setup passes)."""
    t = min(max(step / max_steps, 0.0), 1.0)
    return math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


class OracleTrainer:
    def __init__(self, params, spatial_lr_scale=1.0, position_lr_init=1.6e-4, position_lr_final=1.6e-6,
                 position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=2.5e-3, opacity_lr=0.05,
                 scaling_lr=5e-3, rotation_lr=1e-3, c_init=0.17, c_lr=0.1, sh_degree=3, parallel_views=False):
        self.parallel_views = bool(parallel_views)          # the three renders of step() on three host threads
        P = lambda t: torch.nn.Parameter(t.detach().cpu().clone().float())
        self.p = {k: P(params[k]) for k in ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")}
        self.sched = (position_lr_init * spatial_lr_scale, position_lr_final * spatial_lr_scale, position_lr_delay_mult,
                      position_lr_max_steps)
        groups = [{"params": [self.p["xyz"]], "lr": self.sched[0], "name": "xyz"},
                  {"params": [self.p["features_dc"]], "lr": feature_lr, "name": "f_dc"},
                  {"params": [self.p["features_rest"]], "lr": feature_lr / 20.0, "name": "f_rest"},
                  {"params": [self.p["opacity"]], "lr": opacity_lr, "name": "opacity"},
                  {"params": [self.p["scaling"]], "lr": scaling_lr, "name": "scaling"},
                  {"params": [self.p["rotation"]], "lr": rotation_lr, "name": "rotation"}]
        self.opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15)              # scene/gaussian_model.py:163
        self.c = torch.nn.Parameter(torch.tensor(float(c_init)))
        self.opt_c = torch.optim.Adam([self.c], lr=c_lr)                    # train.py:71-73
        self.sh_degree = sh_degree
        self.iteration = 0

    def activated(self):
        p = self.p
        return (p["xyz"], torch.cat((p["features_dc"], p["features_rest"]), dim=1), torch.sigmoid(p["opacity"]),
                torch.exp(p["scaling"]), torch.nn.functional.normalize(p["rotation"]))

    def render(self, cam, bg):
        with torch.no_grad():
            m, sh, o, s, r = self.activated()
            return _OracleRasterize.apply(m, sh, o, s, r, cam, bg, self.sh_degree)

    def step(self, cam_int, cam_now, cam_next, gt_int, gt_now, gt_next, bg, gt_blur=None):
        self.iteration += 1
        for g in self.opt.param_groups:                                     # update_learning_rate, train.py:97
            if g["name"] == "xyz":
                g["lr"] = expon_lr(self.iteration, *self.sched)
        m, sh, o, s, r = self.activated()
        if self.parallel_views:
            imgs = _OracleRasterizeViews.apply(m, sh, o, s, r, (cam_int, cam_now, cam_next), bg, self.sh_degree)
        else:
            imgs = [_OracleRasterize.apply(m, sh, o, s, r, cam, bg, self.sh_degree) for cam in (cam_int, cam_now, cam_next)]
        loss = torch_oracle.event_iteration_loss(imgs[0], imgs[1], imgs[2], gt_int, gt_now, gt_next, self.c, gt_blur)
        self.opt_c.zero_grad()
        loss.backward()
        self.opt_c.step()
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        return float(loss.detach())
