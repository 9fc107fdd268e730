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
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib, losses, parallel, rasterizer
from .rasterizer import GaussianRasterizationSettings, rasterize_gaussians


class ExponentialLR:
    """The position learning-rate schedule (utils/general_utils.py:29-62 `get_expon_lr_func`; golden G7): geometric
    interpolation lr_init -> lr_final over max_steps, times an optional warm-up factor that eases from delay_mult to 1
    over delay_steps along a quarter sine.  Held as (log lr_init, log ratio), evaluated in float64 with math.*."""

    def __init__(self, lr_init, lr_final, delay_steps=0, delay_mult=1.0, max_steps=1000000):
        self.off = lr_init == 0.0 and lr_final == 0.0
        self.log0 = math.log(lr_init) if lr_init > 0.0 else -math.inf
        self.log1 = math.log(lr_final) if lr_final > 0.0 else -math.inf
        self.delay_steps, self.delay_mult, self.max_steps = delay_steps, delay_mult, max_steps

    def __call__(self, step):
        if step < 0 or self.off:
            return 0.0
        frac = min(max(step / self.max_steps, 0.0), 1.0)
        rate = math.exp(self.log0 * (1.0 - frac) + self.log1 * frac)
        if self.delay_steps > 0:
            ease = math.sin(0.5 * math.pi * min(max(step / self.delay_steps, 0.0), 1.0))
            rate *= self.delay_mult + (1.0 - self.delay_mult) * ease
        return rate


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """The reference's constructor name and keywords (utils/general_utils.py:47) for ExponentialLR."""
    return ExponentialLR(lr_init, lr_final, lr_delay_steps, lr_delay_mult, max_steps)


# (name, floats per Gaussian) in flat-buffer order
SEGMENTS = (("xyz", 3), ("features", 48), ("opacity", 1), ("scaling", 3), ("rotation", 4))
FLOATS_PER_GAUSSIAN = sum(n for _, n in SEGMENTS)   # 59


class EventTrainer:
    """Holds the Gaussian parameters (pre-activation, scene/gaussian_model.py:44-59) and runs steps.

    step()          fused path: no autograd; activations inside the kernels (E3DGS_FLAG_PREACT) and the three
                    backward passes accumulate straight into the flat gradient buffer (E3DGS_FLAG_ACCUMULATE).
    step_autograd() the same iteration through the drop-in autograd operator + torch activations, as the
                    reference's train.py does; kept as the equivalence check of the fused path.
    """

    FWD_FLAGS = _lib.FLAG_PREACT | _lib.FLAG_SH_PLANAR

    def features_reference_layout(self):
        """(N,16,3) view-copy of the SH coefficients in the reference's layout (gaussian_model.py:105-108)."""
        self.sync_features()
        return self.views["features"].t().reshape(self.N, 16, 3).contiguous()

    def __init__(self, params, device, spatial_lr_scale=1.0, position_lr_init=1.6e-4, position_lr_final=1.6e-6,
                 position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=2.5e-3, opacity_lr=0.05,
                 scaling_lr=5e-3, rotation_lr=1e-3, c_init=0.17, c_lr=0.1, active_sh_degree=3, process_group=None,
                 track_densification_stats=False, overlap_features=None, factorize_sh=None, dp_schedule=None,
                 tile_cull=None, small_scene_paths=None, fast_exp=None, force_distributed=False):
        self.device = torch.device(device)
        # Rasteriser options of THIS trainer, carried in the flags word of every call it makes (E3DGS_FLAG_OPTIONS: the
        # library reads no process-wide setting then, so trainers with different options can share a process, also on
        # concurrent streams).  tile_cull / small_scene_paths: None keeps the process defaults (_lib.option_flags).
        # fast_exp (default: environment E3DGS_FAST_EXP=1, else off): TOLERANCE MODE of the compositing kernels -- hardware
        # exp2 instead of the bit-reproducible polynomial; integer binning unchanged, image <= 1e-4 apart from a handful of
        # threshold-flip pixels, gradients <= 1e-3 (tests/test_hip_parity.py::test_fast_exp_mode...).
        if fast_exp is None:
            fast_exp = os.environ.get("E3DGS_FAST_EXP", "0") == "1"
        self.fast_exp = bool(fast_exp)
        self.FWD_FLAGS = _lib.FLAG_PREACT | _lib.FLAG_SH_PLANAR | _lib.option_flags(tile_cull, small_scene_paths, self.fast_exp)
        self.c_lr = c_lr
        self.xyz_lr = get_expon_lr_func(position_lr_init * spatial_lr_scale, position_lr_final * spatial_lr_scale,
                                        lr_delay_mult=position_lr_delay_mult, max_steps=position_lr_max_steps)
        self.lrs = dict(features=feature_lr, features_rest=feature_lr / 20.0, opacity=opacity_lr, scaling=scaling_lr,
                        rotation=rotation_lr)
        self.active_sh_degree = active_sh_degree
        self.iteration = 0
        # torch.optim.Adam keeps one step count PER PARAMETER and skips parameters without a gradient.  In the reference
        # the six Gaussian groups share a count that stalls on densification iterations, opacity additionally stalls
        # on reset iterations (train.py:317-332), and c has its own optimizer that steps on event iterations only
        # (train.py:71-73,210-212).
        self.steps = {"gauss": 0, "opacity": 0, "c": 0}
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0
        # `multi`: the distributed branches of the iteration are taken.  force_distributed (test hook) takes them in a
        # ONE-rank process group too, so that the schedules run over the real RCCL backend on a single GPU.
        self.multi = self.world > 1 or (bool(force_distributed) and dist.is_available() and dist.is_initialized())
        if self.multi and self.world == 1:
            parallel.force_single_rank_collectives(self)       # (for this trainer's lifetime)
        # Exchange of the non-SH groups (11 floats per Gaussian + c) between the ranks:
        #   "allreduce"  in-place mean (RCCL picks ring / tree), every rank then runs Adam on all of them;
        #   "rs_ag"      the direct schedule of SURVEY 5.8: reduce-scatter of the gradients, Adam on the OWNED shard only
        #                (1 / world of the optimizer's HBM traffic and of its moments' updates), all-gather of the updated
        #                parameters.  Same bytes on the links; the replicas stay bit-identical (everyone receives the same
        #                updated shards).  The moments of the other ranks' shards are stale on this rank until
        #                sync_optimizer_state() gathers them (export, densification, checkpoints do).
        sched = dp_schedule or os.environ.get("E3DGS_DP_SCHEDULE") or "allreduce"      # (argument first, as factorize_sh)
        if sched not in ("allreduce", "rs_ag"):
            raise ValueError("dp_schedule must be 'allreduce' or 'rs_ag'")
        self.dp_schedule = sched if self.multi else "allreduce"
        self._shard = None
        self.track_stats = track_densification_stats
        # The SH coefficients are 48 of the 59 floats per Gaussian, and nothing in front of the compositing kernel reads
        # them (E3DGS_FLAG_DEFER_COLOR).  With overlap_features their gradient averaging + Adam run on a second stream
        # and the NEXT iteration's projection / sorts / binning proceed meanwhile; the main stream waits for them right
        # before its colour kernel.  Results are identical to the serial order.  Default: on when there are several
        # ranks (it hides up to ~0.75 ms of the xGMI-bound collective per iteration); off on one GPU, where the
        # HBM-bound Adam and the next preprocess only compete for bandwidth and the separate colour kernel costs
        # 60 us (measured 299 vs 302 iters/s).  E3DGS_OVERLAP=0/1 overrides.
        if overlap_features is None:
            overlap_features = self.multi
        env = os.environ.get("E3DGS_OVERLAP")
        if env is not None:
            overlap_features = env != "0"
        self.overlap_features = bool(overlap_features)
        # Several ranks: the SH gradient (48 of the 59 floats per Gaussian) is not averaged as such.  It is
        # sum_views Y_k(dir) * dL/dcolour, so the ranks all-gather the 3 colour-gradient floats per (Gaussian, view) and
        # their camera centres (9 instead of 48 floats per Gaussian and rank) and each rebuilds the mean SH gradient
        # (e3dgs_sh_grad_from_colour).  E3DGS_FACTORIZE_SH=0 falls back to averaging the SH gradient itself.
        # The choice is made ONCE, at construction, identically on every rank (it decides which collectives a rank
        # issues).  A triplet that mixes frame sizes (utils/camera_utils.py:19-52 sizes every image on its own) hands out
        # the same three-view colour-gradient block (_compute_gradients_two_sizes), so it takes part in either exchange.
        if factorize_sh is None:
            env = os.environ.get("E3DGS_FACTORIZE_SH")
            # by the bytes that cross the links (parallel.sh_exchange_bytes): 9 floats per Gaussian from every other rank
            # against a ring all-reduce of 48 -- the factorised exchange wins while world < 10.67
            factorize_sh = (env != "0") if env is not None else parallel.choose_sh_exchange(self.world) == "factorised"
        self.factorize_sh = self.multi and bool(factorize_sh)
        # One rank: the same factorisation pays inside the GPU.  step() lets the backward hand out the per-view colour
        # gradients (9 floats per Gaussian) instead of the 48-float SH gradient, and ONE streaming kernel rebuilds that
        # gradient in registers and applies Adam to the SH coefficients (e3dgs_sh_adam_from_colour): 0.3 GB less HBM
        # traffic per iteration at 1 M Gaussians, bit-identical parameters.  E3DGS_SH_VIA_COLOUR=0 disables it.
        self.sh_via_colour = (not self.multi) and os.environ.get("E3DGS_SH_VIA_COLOUR", "1") != "0"
        # No host wait inside an iteration: the binning buffers of the forward are sized from the instance counts of the
        # previous iterations (+ margin) and the kernels read the count from device memory
        # (e3dgs_rasterize_forward_multi_capacity); the host looks at the count only after it has enqueued the backward,
        # i.e. while the GPU still has most of the iteration in front of it, and BEFORE it enqueues the optimizer step: a
        # count that did not fit costs a repeated forward / backward, never a wrong update.  E3DGS_NO_HOST_WAIT=0
        # restores begin -> wait -> finish.
        self.no_host_wait = os.environ.get("E3DGS_NO_HOST_WAIT", "1") != "0"
        # render #1 == render #2 of an event iteration (same pose) rendered once: see compute_gradients()
        self.share_coincident_views = os.environ.get("E3DGS_SHARE_VIEWS", "1") != "0"
        # The pixel gradient of a render that only enters the loss through a luminance is rank 1: s(pixel) * w.  The two
        # contrast renders of an event iteration (rgb_to_LUVscale) and the --gray iteration (rgb_to_grayscale) are: the loss
        # kernels hand the scalar field s to the compositing backward, which then runs one colour chain and seven sums per
        # (pixel, entry) instead of three and nine (e3dgs_rasterize_backward_multi_rank1).  E3DGS_RANK1=0: general form.
        self.rank1 = os.environ.get("E3DGS_RANK1", "1") != "0"
        self.shared_pose_iterations = 0
        self._share_stats_on = False
        self._coincide = {}
        self._dstat = None             # render #1's own pixel gradient on shared-pose iterations that collect statistics
        self._instances_per_view = 0.0 # of the last forward whose count is known (_note_count)
        self._capacity = {}            # (N, views, H, W) -> instances the binning buffers are sized for
        self.count_retries = 0         # iterations repeated because the count exceeded the capacity
        self._packed = None            # this rank's [nv x P x 3 colour gradients | nv x 3 camera centres]: a prefix of
        self._gathered = None          # _packed_store (room for three views, allocated once per N)
        self._packed_store = None
        self._gathered_store = None
        self._packed_views = 0         # views whose colour gradients the last backward left in _packed (0: none)
        self._mean_deferred = False    # the last backward left the SH view-direction term out of the xyz gradient
        #                                (FLAG_DEFER_SH_MEAN); _adam_sh_from_colour() adds it inside the SH optimizer kernel
        self._packed_cams = None       # (camera-centre tensors, versions) whose values sit in the tail of _packed
        self._xyz_prev = None          # the means the gradients were computed with (Adam moves them meanwhile)
        self._side = None
        self._feat_event = None
        zeros = lambda t: torch.zeros_like(t)
        groups = {"xyz": params["xyz"], "f_dc": params["features_dc"], "f_rest": params["features_rest"],
                  "opacity": params["opacity"], "scaling": params["scaling"], "rotation": params["rotation"]}
        self._build({k: [v.to(self.device), None, None] for k, v in groups.items()}, c_value=c_init)

    def _build(self, groups, c_value, c_moments=None):
        """(Re)creates the flat buffers from reference-layout tensors  name -> [param, exp_avg, exp_avg_sq]."""
        N = groups["xyz"][0].shape[0]
        # one flat buffer each for parameters / gradients / Adam moments; the last element is the threshold c
        nflat = N * FLOATS_PER_GAUSSIAN + 1
        self.flat = torch.empty(nflat, dtype=torch.float32, device=self.device)
        self.flat_grad = None
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        # features are stored coefficient-major, (16*3, N) (E3DGS_FLAG_SH_PLANAR): rows 0..2 = f_dc, 3..47 = f_rest
        shapes = {"xyz": (N, 3), "features": (48, N), "opacity": (N, 1), "scaling": (N, 3), "rotation": (N, 4)}

        def ref_to_flat(idx):
            feats = torch.cat((groups["f_dc"][idx], groups["f_rest"][idx]), dim=1).reshape(N, 48).t()
            return {"xyz": groups["xyz"][idx], "features": feats, "opacity": groups["opacity"][idx],
                    "scaling": groups["scaling"][idx], "rotation": groups["rotation"][idx]}
        src = ref_to_flat(0)
        has_m = groups["xyz"][1] is not None
        m_src, v_src = (ref_to_flat(1), ref_to_flat(2)) if has_m else (None, None)
        off = 0
        for name, per in SEGMENTS:
            n = N * per
            self.flat[off:off + n].view(shapes[name]).copy_(src[name])
            if has_m:
                self.exp_avg[off:off + n].view(shapes[name]).copy_(m_src[name])
                self.exp_avg_sq[off:off + n].view(shapes[name]).copy_(v_src[name])
            off += n
        self.flat[off] = float(c_value)
        if c_moments is not None:
            self.exp_avg[off] = c_moments[0]; self.exp_avg_sq[off] = c_moments[1]
        self._bind(N)

    def _bind(self, N):
        """Typed views into the flat buffers of N Gaussians + everything else that depends on N."""
        self.N = N
        self._shard = None             # (rs_ag schedule) shard-local optimizer state is rebuilt from the flat buffers
        assert self.flat.numel() == N * FLOATS_PER_GAUSSIAN + 1
        if self.flat_grad is None or self.flat_grad.numel() != self.flat.numel():
            self.flat_grad = torch.zeros_like(self.flat)
        self.views, self.grads, self.seg = {}, {}, {}
        shapes = {"xyz": (N, 3), "features": (48, N), "opacity": (N, 1), "scaling": (N, 3), "rotation": (N, 4)}
        off = 0
        for name, per in SEGMENTS:
            n = N * per
            self.seg[name] = (off, n)
            self.views[name] = self.flat[off:off + n].view(shapes[name])
            self.grads[name] = self.flat_grad[off:off + n].view(shapes[name])
            off += n
        self.seg["c"] = (off, 1)
        self.c = self.flat[off:off + 1]
        self.c_grad = self.flat_grad[off:off + 1]
        # scratch that depends on N
        self.viewspace_grad = torch.zeros(N, 3, dtype=torch.float32, device=self.device) if self.track_stats else None
        self._loss_bufs = None
        # count(D* != 0) of the ground-truth pairs met so far: the event loss sweeps the images once from the second time a
        # pair comes up (losses.PairCounts; E3DGS_EVENT_LOSS_CACHED=0: always the three-launch form)
        self._pair_counts = losses.PairCounts() if os.environ.get("E3DGS_EVENT_LOSS_CACHED", "1") != "0" else None
        self._counts = None
        # scratch of the rasteriser calls of step(): persistent, grows geometrically (no allocator traffic per step);
        # it survives densification (the buffers are sized by bytes, not by N)
        if getattr(self, "_pool", None) is None:
            self._pool = rasterizer.ScratchPool(self.device)
        self.last_radii = None
        self.last_scalars = None

    def export_groups(self):
        """Reference-layout tensors  name -> [param, exp_avg, exp_avg_sq]  (scene/gaussian_model.py:154-163 groups)."""
        self.sync_features()
        if self._shard is not None and self.multi:
            # (rs_ag: the moments of the other ranks' shards are stale here, and gathering them is a COLLECTIVE -- hidden
            # inside an export that typically only rank 0 performs it would deadlock the job)
            raise RuntimeError("dp_schedule='rs_ag': the Adam moments are sharded over the ranks -- call "
                               "sync_optimizer_state() on EVERY rank before export_groups() / checkpointing")
        N = self.N
        out = {}
        for idx, buf in enumerate((self.flat, self.exp_avg, self.exp_avg_sq)):
            def seg(name, shape):
                off, n = self.seg[name]
                return buf[off:off + n].view(shape)
            feats = seg("features", (48, N)).t().reshape(N, 16, 3)
            cur = {"xyz": seg("xyz", (N, 3)).clone(), "f_dc": feats[:, :1].contiguous(), "f_rest": feats[:, 1:].contiguous(),
                   "opacity": seg("opacity", (N, 1)).clone(), "scaling": seg("scaling", (N, 3)).clone(),
                   "rotation": seg("rotation", (N, 4)).clone()}
            for k, v in cur.items():
                out.setdefault(k, [None, None, None])[idx] = v
        return out

    def import_groups(self, groups, steps=None):
        """Rebuild the flat buffers from reference-layout groups (after a torch-side densification, or when resuming
        from a checkpoint).  `steps`: Adam step counts that belong to the imported moments ({"gauss", "opacity"[, "c"]},
        io_formats.restored_steps): without them a resumed trainer would restart the bias correction at step 1 with
        non-zero moments (first updates ~10x too large)."""
        off, _ = self.seg["c"]
        c_val = float(self.c)
        c_mom = (float(self.exp_avg[off]), float(self.exp_avg_sq[off]))
        self._build(groups, c_value=c_val, c_moments=c_mom)
        if steps is not None:
            for k in ("gauss", "opacity", "c"):
                if k in steps:
                    self.steps[k] = int(steps[k])

    def densify_and_prune(self, stats, max_grad=0.0002, min_opacity=0.005, extent=1.0, max_screen_size=None,
                          percent_dense=0.01, sampler=None):
        """scene/gaussian_model.py:389-403 on the device (e3dgs_densify_plan / _apply): one plan pass, one pass that
        compacts parameters and both Adam moments from the old flat buffers into new ones and appends clones and split
        children; no export / import through the reference layout.  `sampler(stds) -> (2 n_split, 3)` draws the split
        offsets (default: the reference's torch.normal(zeros, stds) on the device, :358-360).
        densify.densify_and_prune is the same algorithm in torch on the reference layout (golden-pinned)."""
        import ctypes as C
        self.sync_features()
        self.sync_optimizer_state()
        L = _lib.lib()
        N, dev = self.N, self.device
        scratch = torch.empty(L.e3dgs_densify_scratch_bytes(N), dtype=torch.uint8, device=dev)
        counts = (C.c_int * 4)()
        acc, den = stats.xyz_gradient_accum.contiguous(), stats.denom.contiguous()
        with torch.cuda.device(dev):
            rc = L.e3dgs_densify_plan(N, _lib.ptr(self.flat), _lib.ptr(acc), _lib.ptr(den), float(max_grad),
                                      float(min_opacity), float(extent), float(percent_dense),
                                      1 if max_screen_size else 0, _lib.ptr(scratch), counts, _lib.current_stream())
        _lib.check(rc, "e3dgs_densify_plan")
        n_orig, n_clone, n_split, n_child = (int(c) for c in counts)
        M = n_orig + n_clone + 2 * n_child
        samples = None
        if n_split:
            off = L.e3dgs_densify_split_rows(N, _lib.ptr(scratch)) - scratch.data_ptr()
            rows = scratch[off:off + 4 * n_split].view(torch.int32).long()
            stds = torch.exp(self.views["scaling"].index_select(0, rows)).repeat(2, 1)
            if sampler is None:
                samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=dev), std=stds)
            else:
                samples = sampler(stds).to(device=dev, dtype=torch.float32)
            samples = samples.contiguous()
        new = [torch.empty(M * FLOATS_PER_GAUSSIAN + 1, dtype=torch.float32, device=dev) for _ in range(3)]
        with torch.cuda.device(dev):
            rc = L.e3dgs_densify_apply(N, M, counts, _lib.ptr(self.flat), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
                                       _lib.ptr(samples), _lib.ptr(new[0]), _lib.ptr(new[1]), _lib.ptr(new[2]),
                                       _lib.ptr(scratch), _lib.current_stream())
        _lib.check(rc, "e3dgs_densify_apply")
        self.flat, self.exp_avg, self.exp_avg_sq = new
        self.flat_grad = None
        self._bind(M)
        stats._zero(M, dev)                      # densification_postfix (:329-347) zeroes all three statistics
        return self.N

    def reset_opacity(self):
        """scene/gaussian_model.py:210-213 + replace_tensor_to_optimizer (:258-271): opacity = logit(min(sigmoid, 0.01)),
        its Adam moments zeroed -- in place on the opacity segment (the other groups, their moments and this
        iteration's gradients stay, as in the reference, where only the opacity parameter is replaced)."""
        self.sync_features()
        self.sync_optimizer_state()
        o = self.views["opacity"]
        x = torch.min(torch.sigmoid(o), torch.ones_like(o) * 0.01)
        o.copy_(torch.log(x / (1 - x)))
        off, n = self.seg["opacity"]
        self.exp_avg[off:off + n].zero_()
        self.exp_avg_sq[off:off + n].zero_()

    # ---- raster settings for one view (gaussian_renderer/__init__.py:35-51)
    @staticmethod
    def _camera_tensors(cam):
        """Contiguous copies of the camera's (strided, scene/cameras.py:54-57) matrices, made once per camera
        instead of once per step (three tiny copy kernels per rasteriser call otherwise)."""
        src = (cam.world_view_transform, cam.full_proj_transform, cam.camera_center)
        # the cache entry holds the source tensors themselves and is matched by IDENTITY (+ in-place version): an
        # (address, version) pair does not identify a tensor -- the caching allocator hands a freed block to the next
        # tensor, at the same address, with version 0
        cached = getattr(cam, "_e3dgs_contig", None)
        if cached is None or not EventTrainer._same_tensors(cached[0], src):
            cached = ((src, tuple(t._version for t in src)), tuple(t.contiguous() for t in src))
            try:
                cam._e3dgs_contig = cached
            except AttributeError:       # cameras with __slots__ / namedtuples: no caching
                pass
        return cached[1]

    @staticmethod
    def _same_tensors(entry, tensors):
        """entry = (tensors, versions) of a cache hit candidate: the same tensor OBJECTS, not modified in place since."""
        kept, versions = entry
        return len(kept) == len(tensors) and all(a is b and a._version == v for a, b, v in zip(kept, tensors, versions))

    def _settings(self, cam, bg, scaling_modifier=1.0):
        view, proj, campos = self._camera_tensors(cam)
        return GaussianRasterizationSettings(
            image_height=int(cam.image_height), image_width=int(cam.image_width),
            tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg,
            scale_modifier=scaling_modifier, viewmatrix=view, projmatrix=proj,
            sh_degree=self.active_sh_degree, campos=campos, prefiltered=False, debug=False)

    def render_raw(self, cam, bg):
        """Forward only, fused activations.  Returns the forward_raw dict (image in ["color"])."""
        self.sync_features()
        v = self.views
        return rasterizer.forward_raw(v["xyz"], v["features"], None, v["opacity"], v["scaling"], v["rotation"], None,
                                      self._settings(cam, bg), flags=self.FWD_FLAGS)

    def step(self, cam_int, cam_now, cam_next, gt_int, gt_now, gt_next, bg, gt_blur=None, sync_grads=True):
        """One event iteration (train.py:97-332 without densification).  Returns the device scalars tensor of
        the loss kernel ([0] = loss; a copy the caller owns, as step_image() returns).  The
        three renders are ONE multi-view pass of the rasteriser (every kernel of the pipeline runs once over the three
        cameras), forward and backward.  Nothing waits for the host: forward, loss and backward are enqueued back to back
        with binning buffers sized from earlier instance counts; the host reads this iteration's count once the backward
        is enqueued and only then enqueues the optimizer step (_count_fits).  When cam_int and cam_now are the same view
        (the reference's datasets: compute_gradients) that view is rendered once -- two views per iteration."""
        return self.step_nocopy(cam_int, cam_now, cam_next, gt_int, gt_now, gt_next, bg, gt_blur, sync_grads).clone()

    def step_nocopy(self, cam_int, cam_now, cam_next, gt_int, gt_now, gt_next, bg, gt_blur=None, sync_grads=True):
        """step() without the copy of its result: returns a VIEW of one of two alternating scalar blocks, valid until the
        step after the next one (a loop that only logs the current loss)."""
        scalars = self.compute_gradients(cam_int, cam_now, cam_next, gt_int, gt_now, gt_next, bg, gt_blur,
                                         sh_via_colour=self.sh_via_colour and not self.overlap_features)
        self.apply_update(sync_grads)
        return scalars

    def apply_update(self, sync_grads=True, skip=()):
        """Gradient averaging over the ranks (if any) + Adam (train.py:210-212,330-332) for the iteration whose gradients
        compute_gradients() left in the flat buffer.  `skip` names what torch would leave untouched this iteration:
        "gaussians" (all six groups: a densification iteration or the last iteration), "opacity" (a reset iteration),
        "c" (an iteration without the event loss)."""
        self.iteration += 1
        it = self.iteration
        skip = set(skip)
        if "gaussians" in skip:
            skip.add("opacity")
        st = {}
        for name, key in (("gauss", "gaussians"), ("opacity", "opacity"), ("c", "c")):
            if key in skip:
                st[name] = 0
            else:
                self.steps[name] += 1
                st[name] = self.steps[name]
        dist_on = self.multi and sync_grads
        if self.multi and (not dist_on or "gaussians" in skip):
            self.sync_optimizer_state()        # (rs_ag: these paths update the FULL moment buffers on every rank)
        if "gaussians" in skip:
            if st["c"]:
                if dist_on:
                    parallel.allreduce_mean_(self.c_grad, self.pg)
                so, sn = self.seg["c"]
                self._adam_range(so, sn, self.c_lr, st["c"], eps=1e-8)
            self._packed_views = 0
            self._mean_deferred = False
        elif self.overlap_features or self.factorize_sh:
            self._update_overlapped(it, dist_on, st)
        elif dist_on:
            self._allreduce_and_adam(it, st)                       # 59 floats/Gaussian + c, pipelined with Adam
        elif self._packed_views > 0:
            self._adam_sh_from_colour(it, st)
        else:
            self._adam(it, st)

    def sync_features(self):
        """Make the current stream wait for the SH-coefficient update still running on the side stream (call before
        anything other than step() touches the parameters: rendering, export, densification ...)."""
        if self._feat_event is not None:
            torch.cuda.current_stream(self.device).wait_event(self._feat_event)
            self._feat_event = None

    def _update_overlapped(self, it, dist_on, st):
        """xyz / opacity / scaling / rotation / c: mean over the ranks + Adam on the main stream.  SH coefficients:
        exchange (factorised: all-gather of the per-view colour gradients + local rebuild; otherwise chunked mean of
        the SH gradient) + Adam, on the side stream when overlap_features is set, else on the main stream."""
        main = torch.cuda.current_stream(self.device)
        if self.overlap_features and self._side is None:
            self._side = torch.cuda.Stream(self.device)
        side = self._side if self.overlap_features else main
        chunks = self._comm_chunks()
        small = [c for c in chunks if c[0] != "features"]
        feats = [c for c in chunks if c[0] == "features"]
        fact = self.factorize_sh and self._packed_views > 0
        if fact:
            # the rebuild needs the means the gradients were computed with; Adam (main stream) is about to move them
            if self._xyz_prev is None or self._xyz_prev.shape != self.views["xyz"].shape:
                self._xyz_prev = torch.empty_like(self.views["xyz"])
            self._xyz_prev.copy_(self.views["xyz"])
        grads_ready = main.record_event() if self.overlap_features else None
        mean = (lambda c: parallel.allreduce_mean_async_(self.flat_grad[c[1]:c[1] + c[2]], self.pg)) if dist_on else \
               (lambda c: None)
        # every rank issues its collectives in the same order: small groups first, then the SH exchange
        sharded = dist_on and self.dp_schedule == "rs_ag"
        if not dist_on:
            self.sync_optimizer_state()            # (a local step updates the full moment buffers)
        pend_small = [] if sharded else [(c, mean(c)) for c in small]
        pend_shard = self._nonsh_reduce_scatter() if sharded else None
        with torch.cuda.stream(side):
            if grads_ready is not None:
                side.wait_event(grads_ready)
            if fact:
                gather = parallel.allgather_async_(self._gathered, self._packed, self.pg) if dist_on else None
                pend_feat = None
            else:
                pend_feat = [(c, mean(c)) for c in feats]
        for c, p in pend_small:
            if p is not None:
                p.wait()
            self._adam_chunk(c, it, st)
        if sharded:
            self._nonsh_adam_shard_and_allgather(pend_shard, it, st)
        with torch.cuda.stream(side):
            if fact:
                if gather is not None:
                    packed, nranks = gather.wait(), self.world
                else:                                   # local step: this rank's views only
                    packed, nranks = self._packed.view(1, -1), 1
                # mean SH gradient over all ranks' views rebuilt in registers + Adam of the SH coefficients, one kernel
                f_off, f_n = self.seg["features"]
                rasterizer.sh_adam_from_colour(self._xyz_prev, packed, nranks, self._packed_views, self.active_sh_degree,
                                               16, self.views["features"], self.exp_avg[f_off:f_off + f_n],
                                               self.exp_avg_sq[f_off:f_off + f_n], self.lrs["features"],
                                               self.lrs["features_rest"], st["gauss"], scale=1.0 / nranks)
                self._packed_views = 0
            else:
                for c, p in pend_feat:
                    if p is not None:
                        p.wait()
                    self._adam_chunk(c, it, st)
            if self.overlap_features:
                self._feat_event = side.record_event()

    CAPACITY_MARGIN = 1.25
    CAPACITY_KEYS = 4

    def _forward_views(self, settings, slot=0):
        """The renders of one iteration as ONE multi-view pass.  Without a known capacity (first iteration, new size):
        begin -> host wait for the count -> finish, and the count seeds the capacity.  Afterwards: everything enqueued
        at once with the binning buffers sized by the capacity; _count_fits() is called once the backward is enqueued.
        slot: which scratch pool / count word the pass uses (an iteration that needs TWO passes whose results live side by
        side -- a triplet with two frame sizes -- gives the second one slot 1)."""
        v = self.views
        flags = self.FWD_FLAGS | _lib.FLAG_COUNT_MAPPED
        if self.overlap_features:
            flags |= _lib.FLAG_DEFER_COLOR         # projection / sorts / binning do not read the SH coefficients ...
        if self._counts is None:
            self._counts = torch.zeros(1, dtype=torch.int32).pin_memory()
        counts, pool = self._counts, self._pool
        if slot:
            if getattr(self, "_slot1", None) is None:
                self._slot1 = (torch.zeros(1, dtype=torch.int32).pin_memory(), rasterizer.ScratchPool(self.device))
            counts, pool = self._slot1
        key = (self.N, len(settings), int(settings[0].image_height), int(settings[0].image_width))
        cap = self._capacity.get(key) if self.no_host_wait else None
        if cap is None:
            pend = rasterizer.forward_multi_begin(v["xyz"], v["features"], v["opacity"], v["scaling"], v["rotation"],
                                                  settings, flags=flags, count_host=counts, pool=pool)
            pend.before_colour = self.sync_features    # ... whose update (side stream) must be done before the colour kernel
            rasterizer.prepare_multi_finish(pend)      # (host work done while the GPU still computes the count)
            rasterizer.wait_count(pend)                # the host wait: the instance count (polled)
            raw = rasterizer.forward_multi_finish(pend)
            self._note_count(key, raw["num_rendered"])
            return raw
        raw = rasterizer.forward_multi_capacity(v["xyz"], v["features"], v["opacity"], v["scaling"], v["rotation"], settings,
                                                cap, counts, flags=flags, pool=pool,
                                                before_colour=self.sync_features if self.overlap_features else None)
        raw["capacity_key"] = key
        return raw

    SHARE_STATS_MIN_INSTANCES = 500_000      # per view, and
    SHARE_STATS_MIN_TILES = 4096             # tiles per view: below either, statistics iterations keep three renders

    def _note_count(self, key, count):
        self._instances_per_view = count / max(int(key[1]), 1)
        if not self.no_host_wait:
            return
        cap = self._capacity.get(key)
        if cap is None or count > 0.95 * cap:          # (stable otherwise: the scratch pool and its pointers stay put)
            self._capacity.pop(key, None)
            while len(self._capacity) >= self.CAPACITY_KEYS:         # the most recent few (views, frame size) combinations:
                self._capacity.pop(next(iter(self._capacity)))       # 2- and 3-view iterations alternate on real datasets
            self._capacity[key] = int(count * self.CAPACITY_MARGIN) + 65536

    def _count_fits(self, raw):
        """For a pre-sized forward: wait for its instance count (the GPU produced it early in the iteration; by now it is
        busy with the kernels enqueued since) and say whether the buffers were large enough."""
        if "capacity" not in raw:
            return True
        count = rasterizer.wait_count(raw)
        self._note_count(raw["capacity_key"], count)
        if count > raw["capacity"]:
            self.count_retries += 1
            return False
        return True

    def compute_gradients(self, cam_int, cam_now, cam_next, gt_int, gt_now, gt_next, bg, gt_blur=None, sh_via_colour=False,
                          viewspace_grad=True):
        """Forward (three renders), event loss and backward of one iteration: fills the flat gradient buffer.
        sh_via_colour (what step() uses on one rank): the SH segment of the gradient buffer is NOT written; the backward
        leaves the per-view colour gradients instead and apply_update() rebuilds the SH gradient inside the fused
        SH-optimizer kernel.
        Shared pose (share_coincident_views, on by default): when cam_int and cam_now are the same view -- the
        reference's datasets are built that way: the event cameras are read with the training cameras' extrinsics
        (scene/dataset_readers.py:157), so train.py:144 and :159 render the same image twice -- it is rendered ONCE: its
        image feeds the intensity and the contrast term and its backward receives the sum of the two pixel gradients
        (the backward is linear in them).  Same loss, same image bits, gradients equal to summation order, two thirds of
        the work.  When the densification statistics are collected (track_densification_stats, train.py:145,317-320) they
        need the screen-space gradient of render #1 ALONE: the shared view's tiles then run a second dL/dalpha chain on
        render #1's own pixel gradient (e3dgs_rasterize_backward_multi_stats); viewspace_grad=False says the statistics
        are not updated this iteration (fit.fit_event_scene passes the schedule's answer) and saves that chain.
        The returned scalars tensor, `last_scalars` and `last_radii` are VIEWS of persistent buffers that the next
        iteration overwrites (step() / step_image() return clones)."""
        v = self.views
        # ground-truth frames as plain fp32 (3, H, W) planes (a no-op for frames that already are: cameras of the
        # reference hold whatever strides PILtoTorch's permute left, utils/general_utils.py:21-27)
        plane = lambda t: None if t is None else (t if (t.dtype == torch.float32 and t.is_contiguous())
                                                  else t.to(torch.float32).contiguous())
        gt_int, gt_now, gt_next, gt_blur = plane(gt_int), plane(gt_now), plane(gt_next), plane(gt_blur)
        settings = [self._settings(c, bg) for c in (cam_int, cam_now, cam_next)]
        sizes = {(int(s.image_height), int(s.image_width)) for s in settings}
        need_vs = self.track_stats and viewspace_grad
        if len(sizes) != 1:
            # the reference sizes every image on its own (utils/camera_utils.py:19-52): a triplet may mix resolutions
            return self._compute_gradients_two_sizes(settings, gt_int, gt_now, gt_next, gt_blur, sh_via_colour, need_vs)
        shared = self.share_coincident_views and self._views_coincide(cam_int, cam_now, settings)
        if shared and need_vs:
            # launches that do not fill the GPU several times over gain nothing: a third view composites beside the other
            # two for almost nothing, while the second gradient chain lengthens the tiles that decide the launch's
            # duration (800 x 800: 30 k Gaussians 0.53 ms with three renders, 0.55 ms shared + statistics, 200 k: 0.95 /
            # 0.91; 1080p: 500 k 1.96 / 1.74, 1 M 2.52 / 2.25 -- tools/shared_pose_time.py)
            tiles = ((settings[0].image_width + 15) // 16) * ((settings[0].image_height + 15) // 16)
            # (hysteresis: the decision only flips when the instance count leaves a +-10 % band around the threshold, so a
            # scene that hovers there does not alternate between two- and three-view iterations)
            lim = self.SHARE_STATS_MIN_INSTANCES * (0.9 if self._share_stats_on else 1.1)
            self._share_stats_on = tiles >= self.SHARE_STATS_MIN_TILES and self._instances_per_view >= lim
            if not self._share_stats_on:
                shared = False
        if shared:
            settings = [settings[0], settings[2]]
        self.shared_pose_iterations += int(shared)
        for _attempt in range(4):
            scalars, raw = self._event_forward_backward(settings, gt_int, gt_now, gt_next, gt_blur, sh_via_colour,
                                                        shared=shared, viewspace=need_vs)
            if self._count_fits(raw):
                break
        else:
            raise RuntimeError("the instance count kept outgrowing the binning capacity")
        self.last_radii = raw["radii"][0]
        self.last_scalars = scalars
        return scalars

    def _views_coincide(self, cam_a, cam_b, settings):
        """True when the two cameras are the same view (frame size, field of view, matrices, centre -- compared by value
        ONCE per camera pair: the answer is cached on the pair's matrix tensors, matched by identity like _camera_tensors)."""
        if cam_a is cam_b:
            return True
        sa, sb = settings[0], settings[1]
        if (sa.image_height, sa.image_width, sa.tanfovx, sa.tanfovy) != (sb.image_height, sb.image_width, sb.tanfovx, sb.tanfovy):
            return False
        tensors = (sa.viewmatrix, sa.projmatrix, sa.campos, sb.viewmatrix, sb.projmatrix, sb.campos)
        # (the entry holds the six small matrix tensors -- never the cameras, which carry their frames -- and is matched by
        # identity + in-place version, as in _camera_tensors)
        key = tuple(id(t) for t in tensors)
        hit = self._coincide.get(key)
        if hit is not None and self._same_tensors(hit[0], tensors):
            return hit[1]
        same = bool(torch.equal(sa.viewmatrix, sb.viewmatrix) and torch.equal(sa.projmatrix, sb.projmatrix)
                    and torch.equal(sa.campos, sb.campos))          # (a host wait, once per camera pair)
        if len(self._coincide) > 4096:
            self._coincide.clear()
        self._coincide[key] = ((tensors, tuple(t._version for t in tensors)), same)
        return same

    def _backward_flags(self, raw, sh_via_colour):
        """One rank, SH gradient never stored (sh_via_colour): the backward also leaves out the part of the xyz gradient that
        comes through the SH view directions -- it needs all 48 coefficients, which the SH optimizer kernel streams over
        anyway (apply_update() -> _adam_sh_from_colour adds it, bit-identical).  Until then grads["xyz"] is incomplete."""
        self._mean_deferred = bool(sh_via_colour and not self.multi and not self.factorize_sh and not self.overlap_features
                                   and self._packed_views <= 4 and os.environ.get("E3DGS_DEFER_SH_MEAN", "1") != "0")
        return raw["flags"] | (_lib.FLAG_DEFER_SH_MEAN if self._mean_deferred else 0)

    def _colour_gradients_instead_of_sh(self, out, settings, pad_to=None):
        """The backward hands out the per-view colour gradients (3 floats per Gaussian and view) instead of the 48-float SH
        gradient, which is rebuilt after the exchange / inside the SH optimizer kernel.  _packed = [nv x P x 3 colour
        gradients | nv x 3 camera centres].  pad_to: the block keeps that many views (the extra ones carry zero
        gradients, which the rebuild skips) -- every rank's block of the all-gather has the same size even when this
        rank rendered a shared-pose iteration with two views."""
        n_real, P = len(settings), self.N
        nv = max(n_real, pad_to or 0)
        # (storage for three views, allocated once: iterations with one, two and three views alternate on real datasets and
        # must not reallocate 9 P floats -- and the gather buffer -- each time; the active block is a prefix)
        need = max(nv, 3) * (P * 3 + 3)
        if self._packed_store is None or self._packed_store.numel() != need:
            self._packed_store = torch.empty(need, dtype=torch.float32, device=self.device)
            self._gathered_store = torch.empty(self.world, need, dtype=torch.float32, device=self.device) \
                if self.multi else None
            self._packed_cams = None
        if self._packed is None or self._packed.numel() != nv * (P * 3 + 3) or \
                self._packed.data_ptr() != self._packed_store.data_ptr():
            self._packed = self._packed_store[:nv * (P * 3 + 3)]
            self._gathered = None if self._gathered_store is None else \
                self._gathered_store.view(-1)[:self.world * self._packed.numel()].view(self.world, -1)
            self._packed_cams = None
        self._packed_views = nv
        out["sh"] = None
        out["colour_views"] = self._packed[:n_real * P * 3].view(n_real, P, 3)
        if nv > n_real:
            self._packed[n_real * P * 3:nv * P * 3].zero_()
        tail = self._packed[nv * P * 3:].view(nv, 3)
        # (same camera-centre tensors as last iteration, unmodified: already there.  The entry keeps the tensors alive
        # and is matched by identity: a stale hit would rebuild the SH gradient with last iteration's directions)
        cams = tuple(st.campos for st in settings) + (settings[-1].campos,) * (nv - n_real)
        if self._packed_cams is None or not self._same_tensors(self._packed_cams, cams):
            # (one launch for the three centres: a 12-byte tensor.copy_ is a hipMemcpyAsync each -- a blit kernel plus its
            # barrier packets, ~12 us of stream time apiece, and a random camera per iteration, train.py:116-131, pays
            # them every iteration: bench.py trained_random_camera measured +40 us per iteration against a fixed triplet)
            torch._foreach_copy_([tail[k] for k in range(len(cams))], [t.reshape(3) for t in cams])
            self._packed_cams = (cams, tuple(t._version for t in cams))

    def _event_forward_backward(self, settings, gt_int, gt_now, gt_next, gt_blur, sh_via_colour, shared=False,
                                viewspace=True):
        # ---- the three renders (train.py:144,159,161); shared: render #1 and render #2 are the same render (two views:
        # [shared, next]) -- its image feeds both loss terms and receives the sum of their gradients
        raw = self._forward_views(settings)
        imgs = raw["color"]
        key = ("event",) + tuple(imgs.shape)
        if self._loss_bufs is None or self._loss_bufs[0] != key:
            # (two scalar blocks used alternately: the block an iteration returns stays intact during the next one)
            self._loss_bufs = (key, torch.empty(2, 8, dtype=torch.float32, device=self.device), torch.empty_like(imgs),
                               torch.empty(_lib.lib().e3dgs_event_loss_scratch_bytes(imgs.shape[3], imgs.shape[2]),
                                           dtype=torch.uint8, device=self.device))
        _, sc2, dpix, scratch = self._loss_bufs
        self._loss_flip = 1 - getattr(self, "_loss_flip", 0)
        sc = sc2[self._loss_flip]
        # (dL/dc goes straight into the threshold's slot of the flat gradient buffer: no copy kernel)
        i_now, i_next = (0, 1) if shared else (1, 2)
        want_vs = self.track_stats and viewspace
        d_int = dpix[0]
        if shared and want_vs:
            # the statistics need render #1's OWN screen-space gradient: its pixel gradient is kept apart (d_int), the
            # shared view's backward gets the sum, and a second dL/dalpha chain in the tiles of view 0 carries d_int
            # (e3dgs_rasterize_backward_multi_stats)
            if self._dstat is None or self._dstat.shape != dpix[0].shape:
                self._dstat = torch.empty_like(dpix[0])
            d_int = self._dstat
        scalars, _, _, _ = losses.event_loss_raw(imgs[0], imgs[i_now], imgs[i_next], self.c, gt_int, gt_now, gt_next, gt_blur,
                                                 out=(sc, d_int, dpix[i_now], dpix[i_next], scratch),
                                                 dc_out=self.c_grad, pair_counts=self._pair_counts,
                                                 rank1=self.rank1)                                                     # train.py:165-203
        # (rank 1: the contrast renders that are renders of their own -- `next`, and `now` unless it shares render #1's pose)
        r1 = None
        if self.rank1:
            r1 = {i_next: rasterizer.LUV_WEIGHTS}
            if not shared:
                r1[i_now] = rasterizer.LUV_WEIGHTS
        # (image and img_now are the same tensor, the outputs are not: the kernel stored the sum in dpix[0], d_int alone)
        # ---- loss.backward() (train.py:211): every gradient element is written exactly once
        g = self.grads
        out = dict(means3D=g["xyz"], sh=g["features"], opacities=g["opacity"], scales=g["scaling"], rots=g["rotation"])
        if self.factorize_sh or sh_via_colour:
            # (several ranks: the block always holds three views -- a rank that rendered a shared-pose iteration pads with
            # zero gradients -- so the all-gather blocks keep one size)
            self._colour_gradients_instead_of_sh(out, settings, pad_to=3 if self.multi else None)
        if want_vs:
            out["means2D"] = self.viewspace_grad            # densification statistics use render #1 only (train.py:145)
        rasterizer.backward_multi(raw, dpix, out, flags=self._backward_flags(raw, sh_via_colour),
                                  stats_grad_view0=d_int if (shared and want_vs) else None, rank1=r1)
        return scalars, raw

    def _compute_gradients_two_sizes(self, settings, gt_int, gt_now, gt_next, gt_blur, sh_via_colour, need_vs):
        """The same iteration for a triplet whose intensity frame has another size than the event pair (the contrast is a
        per-pixel difference, utils/loss_utils.py:234-249: the two event frames must agree): TWO multi-view passes --
        [intensity] and [now, next] -- each with the whole fused pipeline (in-kernel activations, rank-1 contrast gradients,
        per-view colour gradients instead of the SH gradient where the caller takes that route, no host wait), their
        non-SH gradients added.  Everything downstream -- the SH optimizer kernel on one rank, the factorised exchange
        under DP -- sees the three views' colour-gradient block it always sees, so a dataset that mixes resolutions keeps
        the fused path and the collectives of every other rank.  Same mathematics as train.py:165-203:
        loss = 0.9 L1(contrast) rho + 0.1 L1(intensity) (1 - rho)  [+ deblur]."""
        if (settings[1].image_height, settings[1].image_width) != (settings[2].image_height, settings[2].image_width):
            raise ValueError("the two event frames of an iteration must have the same size")
        f32 = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()
        g = self.grads
        via_colour = self.factorize_sh or sh_via_colour
        # second gradient buffer (pass [now, next]): the non-SH segments, packed; the 48 N floats of the SH segment only when
        # its gradient goes to memory at all (via_colour: never written -- 192 MB at 1 M Gaussians that nobody would touch)
        names2 = [n_ for n_ in ("xyz", "opacity", "scaling", "rotation") + (() if via_colour else ("features",))]
        need2 = sum(self.seg[n_][1] for n_ in names2)
        if getattr(self, "_grad2", None) is None or self._grad2.numel() != need2:
            self._grad2 = torch.empty(need2, dtype=torch.float32, device=self.device)
        g2, o2 = {}, 0
        for n_ in names2:
            cnt = self.seg[n_][1]
            g2[n_] = self._grad2[o2:o2 + cnt].view(self.grads[n_].shape)
            o2 += cnt
        g2.setdefault("features", None)
        for _attempt in range(4):
            raw_i = self._forward_views(settings[:1], slot=1)
            raw_e = self._forward_views(settings[1:], slot=0)
            img, now, nxt = raw_i["color"][0], raw_e["color"][0], raw_e["color"][1]
            # contrast term with the event-loss kernel: the intensity slot is fed a frame of the PAIR's size as render and
            # target (L1 = 0 there, no gradient) -> rho, L1(contrast), dL/dc, and the two contrast renders' pixel gradients
            # (loss buffers kept per pair of frame sizes, as _loss_bufs of the uniform path: nothing is allocated per iteration)
            key2 = ("event2",) + tuple(raw_e["color"].shape) + tuple(img.shape)
            if getattr(self, "_loss_bufs2", None) is None or self._loss_bufs2[0] != key2:
                self._loss_bufs2 = (key2, torch.zeros_like(raw_e["color"]), torch.empty(8, device=self.device),
                                    torch.empty_like(now),
                                    torch.empty(_lib.lib().e3dgs_event_loss_scratch_bytes(now.shape[2], now.shape[1]),
                                                dtype=torch.uint8, device=self.device),
                                    torch.zeros(2, 8, dtype=torch.float32, device=self.device))
            _, dpix_e, sc_buf, d_dummy, scr2, scal2 = self._loss_bufs2
            out_l = (sc_buf, d_dummy, dpix_e[0], dpix_e[1], scr2)
            sc, _, _, _ = losses.event_loss_raw(f32(gt_now), now, nxt, self.c, f32(gt_now), f32(gt_now), f32(gt_next), None,
                                                out=out_l, pair_counts=self._pair_counts, rank1=self.rank1)
            rho = sc[2]
            e_int = img - f32(gt_int)
            l1_int = e_int.abs().mean()
            d_img = (0.1 * (1.0 - rho) / e_int.numel()) * torch.sign(e_int)
            loss = sc[0] + 0.1 * l1_int * (1.0 - rho)
            dc = sc[1]
            l1_blur = None
            if gt_blur is not None:                                            # train.py:197-203
                e_b = img - f32(gt_blur)
                l1_blur = e_b.abs().mean()
                loss = 0.5 * loss + 0.5 * l1_blur
                d_img = 0.5 * d_img + (0.5 / e_b.numel()) * torch.sign(e_b)
                dpix_e.mul_(0.5)
                dc = 0.5 * dc
            self._loss_flip2 = 1 - getattr(self, "_loss_flip2", 0)
            scalars = scal2[self._loss_flip2]             # (two blocks used alternately, as the uniform path's)
            scalars.zero_()
            scalars[0], scalars[1], scalars[2], scalars[3], scalars[4] = loss, dc, rho, sc[3], l1_int
            if l1_blur is not None:
                scalars[5] = l1_blur                      # (the uniform path reports the blur L1 there too)
            # ---- backward: pass [intensity] into the gradient buffer, pass [now, next] into a second one
            out_i = dict(means3D=g["xyz"], sh=g["features"], opacities=g["opacity"], scales=g["scaling"], rots=g["rotation"])
            out_e = dict(means3D=g2["xyz"], sh=g2["features"], opacities=g2["opacity"], scales=g2["scaling"], rots=g2["rotation"])
            if via_colour:
                both = dict(out_i)
                self._colour_gradients_instead_of_sh(both, settings, pad_to=3 if self.multi else None)
                cv = both["colour_views"]
                out_i["sh"] = out_e["sh"] = None
                out_i["colour_views"], out_e["colour_views"] = cv[0:1], cv[1:3]
            else:
                self._packed_views = 0
            if need_vs:
                out_i["means2D"] = self.viewspace_grad              # densification statistics use render #1 only (train.py:145)
            rasterizer.backward_multi(raw_i, d_img.contiguous()[None], out_i, flags=self._backward_flags(raw_i, sh_via_colour))
            r1 = {0: rasterizer.LUV_WEIGHTS, 1: rasterizer.LUV_WEIGHTS} if self.rank1 else None
            rasterizer.backward_multi(raw_e, dpix_e, out_e, flags=self._backward_flags(raw_e, sh_via_colour), rank1=r1)
            fits = [self._count_fits(raw_i), self._count_fits(raw_e)]
            if all(fits):
                break
        else:
            raise RuntimeError("the instance count kept outgrowing the binning capacity")
        # sum of the two passes (the SH segment only when its gradient is in memory)
        for n_ in names2:
            self.grads[n_].add_(g2[n_])
        self.c_grad.copy_(scalars[1:2])
        self.last_radii = raw_i["radii"][0]
        self.last_scalars = scalars
        return scalars

    # ---- multi-GPU: the gradient buffer is averaged in 6 contiguous chunks (xyz | 4 quarters of the features |
    # opacity+scaling+rotation+c); the chunks' collectives are issued back to back and the Adam launches of chunk k
    # wait only for chunk k, so the HBM-bound optimizer hides under the xGMI-bound collectives that follow it
    FEATURE_CHUNKS = 4

    def _comm_chunks(self):
        N = self.N
        f_off, f_n = self.seg["features"]
        q = (f_n // self.FEATURE_CHUNKS + 255) // 256 * 256
        chunks = [("xyz", self.seg["xyz"][0], self.seg["xyz"][1])]
        o = 0
        while o < f_n:
            chunks.append(("features", f_off + o, min(q, f_n - o)))
            o += q
        t_off = self.seg["opacity"][0]
        chunks.append(("tail", t_off, self.flat.numel() - t_off))
        return chunks

    def _allreduce_and_adam(self, it, st):
        chunks = self._comm_chunks()
        sharded = self.dp_schedule == "rs_ag"
        if sharded:                                   # non-SH groups: reduce-scatter -> Adam on the shard -> all-gather
            pend_shard = self._nonsh_reduce_scatter()
            chunks = [c for c in chunks if c[0] == "features"]
        pend = [parallel.allreduce_mean_async_(self.flat_grad[off:off + n], self.pg) for _, off, n in chunks]
        if sharded:
            self._nonsh_adam_shard_and_allgather(pend_shard, it, st)
        for c, p in zip(chunks, pend):
            p.wait()
            self._adam_chunk(c, it, st)

    # ---- direct reduce-scatter + all-gather schedule for the non-SH groups (dp_schedule == "rs_ag")
    # staging order: xyz (3N) | opacity (N) | scaling (3N) | rotation (4N) | c (1), padded to world * shard elements
    def _nonsh_regions(self):
        t_off = self.seg["opacity"][0]
        return (self.seg["xyz"], (t_off, self.flat.numel() - t_off))

    def _ensure_shards(self):
        if self._shard is not None:
            return self._shard
        n = sum(r[1] for r in self._nonsh_regions())
        sh = (n + self.world - 1) // self.world
        z = lambda k: torch.zeros(k, dtype=torch.float32, device=self.device)
        S = dict(n=n, size=sh, g=z(self.world * sh), p=z(self.world * sh), gs=z(sh), ps=z(sh), m=z(sh), v=z(sh))
        for buf, dst in ((self.exp_avg, S["m"]), (self.exp_avg_sq, S["v"])):       # this rank's slice of the moments
            self._stage(buf, S["p"])
            dst.copy_(S["p"][self.rank * sh:(self.rank + 1) * sh])
        self._shard = S
        return S

    def _stage(self, flat_buf, staging):
        o = 0
        for off, n in self._nonsh_regions():
            staging[o:o + n].copy_(flat_buf[off:off + n])
            o += n

    def _unstage(self, staging, flat_buf):
        o = 0
        for off, n in self._nonsh_regions():
            flat_buf[off:off + n].copy_(staging[o:o + n])
            o += n

    def _nonsh_reduce_scatter(self):
        S = self._ensure_shards()
        self._stage(self.flat_grad, S["g"])
        return parallel.reduce_scatter_mean_async_(S["gs"], S["g"], self.pg)

    def _nonsh_adam_shard_and_allgather(self, pend, it, st):
        S = self._shard
        sh, n, N = S["size"], S["n"], self.N
        a = self.rank * sh
        b = min(a + sh, n)
        self._stage(self.flat, S["p"])
        S["ps"].copy_(S["p"][a:a + sh])
        pend.wait()                                       # mean gradient of the owned shard
        if b > a:
            ends_abs = (3 * N, 4 * N, 7 * N, 11 * N, 11 * N + 1)                   # xyz | opacity | scaling | rotation | c
            ends = tuple(min(max(e - a, 0), b - a) for e in ends_abs)
            lrs = (self.xyz_lr(it), self.lrs["opacity"], self.lrs["scaling"], self.lrs["rotation"], self.c_lr)
            g, o, c = st["gauss"], st["opacity"], st["c"]
            k = b - a
            losses.adam_step_segments_(S["ps"][:k], S["gs"][:k], S["m"][:k], S["v"][:k], ends, lrs,
                                       (1e-15,) * 4 + (1e-8,), (g, o, g, g, c))
        parallel.allgather_flat_async_(S["p"], S["ps"], self.pg).wait()
        self._unstage(S["p"], self.flat)

    def sync_optimizer_state(self):
        """dp_schedule == "rs_ag": gather the moments every rank keeps for its own shard of the non-SH groups into the
        full exp_avg / exp_avg_sq buffers (collective: every rank calls it -- before exporting, checkpointing, densifying
        or resetting).  A no-op for the other schedules."""
        S = self._shard
        if S is None or not self.multi:
            return
        for shard, full in ((S["m"], self.exp_avg), (S["v"], self.exp_avg_sq)):
            parallel.allgather_flat_async_(S["p"], shard, self.pg).wait()
            self._unstage(S["p"], full)
        self._shard = None                                 # rebuilt from the (now complete) flat buffers when next needed

    def _adam_chunk(self, chunk, it, st):
        kind, off, n = chunk
        f_off, _ = self.seg["features"]
        if kind == "xyz":
            self._adam_range(off, n, self.xyz_lr(it), st["gauss"])
        elif kind == "features":
            # f_dc rows (the first 3N elements of the segment) use feature_lr, everything after feature_lr / 20
            dc_left = max(0, f_off + 3 * self.N - off)
            if dc_left > 0:
                self._adam_range(off, n, self.lrs["features"], st["gauss"], lr_b=self.lrs["features_rest"], period=n,
                                 split=min(dc_left, n))
            else:
                self._adam_range(off, n, self.lrs["features_rest"], st["gauss"])
        else:
            for name in ("opacity", "scaling", "rotation"):
                so, sn = self.seg[name]
                self._adam_range(so, sn, self.lrs[name], st["opacity" if name == "opacity" else "gauss"])
            so, sn = self.seg["c"]
            self._adam_range(so, sn, self.c_lr, st["c"], eps=1e-8)

    def _adam_range(self, off, n, lr, step, eps=1e-15, **kw):
        if step <= 0:               # a group without a gradient this iteration: torch leaves it untouched
            return
        sl = slice(off, off + n)
        losses.adam_step_(self.flat[sl], self.flat_grad[sl], self.exp_avg[sl], self.exp_avg_sq[sl], lr, step, eps=eps, **kw)

    def _adam_sh_from_colour(self, it, st):
        """One rank, compute_gradients(sh_via_colour=True): SH gradient rebuild + SH Adam in one streaming kernel (it
        needs the positions the gradients were computed with, so it runs first), then the other groups."""
        N = self.N
        f_off, f_n = self.seg["features"]
        rasterizer.sh_adam_from_colour(self.views["xyz"], self._packed.view(1, -1), 1, self._packed_views,
                                       self.active_sh_degree, 16, self.views["features"],
                                       self.exp_avg[f_off:f_off + f_n], self.exp_avg_sq[f_off:f_off + f_n],
                                       self.lrs["features"], self.lrs["features_rest"], st["gauss"],
                                       mean_grad=self.grads["xyz"] if self._mean_deferred else None)
        self._packed_views = 0
        self._mean_deferred = False
        g, o, c = st["gauss"], st["opacity"], st["c"]
        if os.environ.get("E3DGS_ADAM_GAP", "1") != "0":
            # everything around the SH segment in ONE launch (the segment is a gap the launch does not visit): xyz, then
            # opacity | scaling | rotation | c
            ends = (f_off + f_n,) + tuple(sum(self.seg[n]) for n in ("opacity", "scaling", "rotation", "c"))
            lrs = (self.xyz_lr(it), self.lrs["opacity"], self.lrs["scaling"], self.lrs["rotation"], self.c_lr)
            losses.adam_step_segments_(self.flat, self.flat_grad, self.exp_avg, self.exp_avg_sq, ends, lrs,
                                       (1e-15,) * 4 + (1e-8,), (g, o, g, g, c), gap=(f_off, f_n))
            return
        sl = slice(0, f_off)                                     # xyz (A/B switch: one launch per contiguous range)
        losses.adam_step_segments_(self.flat[sl], self.flat_grad[sl], self.exp_avg[sl], self.exp_avg_sq[sl], (f_off,),
                                   (self.xyz_lr(it),), (1e-15,), st["gauss"])
        t0 = f_off + f_n                                         # opacity | scaling | rotation | c
        sl = slice(t0, self.flat.numel())
        ends = tuple(sum(self.seg[n]) - t0 for n in ("opacity", "scaling", "rotation", "c"))
        lrs = (self.lrs["opacity"], self.lrs["scaling"], self.lrs["rotation"], self.c_lr)
        losses.adam_step_segments_(self.flat[sl], self.flat_grad[sl], self.exp_avg[sl], self.exp_avg_sq[sl], ends, lrs,
                                   (1e-15,) * 3 + (1e-8,), (o, g, g, c))

    def _adam(self, it, st=None):
        """All groups in one launch: the flat buffer is xyz | f_dc | f_rest | opacity | scaling | rotation | c."""
        N = self.N
        f_off, f_n = self.seg["features"]
        ends = (self.seg["xyz"][0] + self.seg["xyz"][1], f_off + 3 * N, f_off + f_n,
                sum(self.seg["opacity"]), sum(self.seg["scaling"]), sum(self.seg["rotation"]), sum(self.seg["c"]))
        lrs = (self.xyz_lr(it), self.lrs["features"], self.lrs["features_rest"], self.lrs["opacity"], self.lrs["scaling"],
               self.lrs["rotation"], self.c_lr)
        eps = (1e-15,) * 6 + (1e-8,)           # scene/gaussian_model.py:163; torch.optim.Adam([c], lr=0.1) train.py:73
        if st is None:
            steps = it
        else:
            g, o, c = st["gauss"], st["opacity"], st["c"]
            steps = g if (g == o == c) else (g, g, g, o, g, g, c)
        losses.adam_step_segments_(self.flat, self.flat_grad, self.exp_avg, self.exp_avg_sq, ends, lrs, eps, steps)

    # ------------------------------------------------------------------ the other two training modes of train.py
    def step_image(self, cam, gt_image, bg, mode="gray", lambda_dssim=0.2, sync_grads=True):
        """One iteration of the reference's `--gray` (train.py:213-223) or RGB (train.py:292-296) mode on the fused
        path: ONE render, loss = (1 - lambda) L1 + lambda (1 - SSIM) (gray: both terms on rgb_to_grayscale), backward,
        Adam.  No autograd graph; the SSIM kernel returns its own gradient.  Returns the loss as a device scalar."""
        if mode not in ("gray", "rgb"):
            raise ValueError("mode must be 'gray' or 'rgb'")
        loss = self.compute_gradients_image(cam, gt_image, bg, mode, lambda_dssim,
                                            sh_via_colour=self.sh_via_colour and not self.overlap_features)
        self.apply_update(sync_grads, skip=("c",))     # optimizer_c only steps on event iterations (train.py:210-212)
        return loss.clone()

    def compute_gradients_image(self, cam, gt_image, bg, mode="gray", lambda_dssim=0.2, sh_via_colour=False):
        """sh_via_colour (what step_image() uses on one rank): as in compute_gradients() -- the SH segment of the gradient
        buffer is NOT written; apply_update() rebuilds the SH gradient inside the SH optimizer kernel."""
        settings = [self._settings(cam, bg)]
        for _attempt in range(4):
            loss, raw = self._image_forward_backward(settings, gt_image, mode, lambda_dssim, sh_via_colour)
            if self._count_fits(raw):
                break
        else:
            raise RuntimeError("the instance count kept outgrowing the binning capacity")
        self.c_grad.zero_()                        # the contrast threshold only exists in the event loss
        self.last_radii = raw["radii"][0]
        self.last_scalars = loss
        return loss

    def _image_forward_backward(self, settings, gt_image, mode, lambda_dssim, sh_via_colour=False):
        raw = self._forward_views(settings)
        img = raw["color"][0]
        gt = gt_image if gt_image.dtype == torch.float32 else gt_image.float()
        gt = gt.contiguous()
        C, H, W = img.shape
        key = ("img", C, H, W)
        if self._loss_bufs is None or self._loss_bufs[0] != key:
            self._loss_bufs = (key, torch.empty(4, dtype=torch.float32, device=self.device),
                               torch.empty(1, C, H, W, device=self.device),
                               torch.empty(_lib.lib().e3dgs_image_loss_scratch_bytes(C, H, W), dtype=torch.uint8,
                                           device=self.device))
        _, sc, dpix, scratch = self._loss_bufs
        # (1 - lambda) L1 + lambda (1 - SSIM) and its image gradient, fused (gray: both terms on rgb_to_grayscale,
        # utils/loss_utils.py:18-23,40-48,368-385; RGB: :270-271,388-396)
        r1 = self.rank1 and mode == "gray" and C == 3
        losses.image_loss_raw(img, gt, mode == "gray", lambda_dssim, out=(sc, dpix[0], scratch), rank1=r1)
        loss = sc[0]
        g = self.grads
        out = dict(means3D=g["xyz"], sh=g["features"], opacities=g["opacity"], scales=g["scaling"], rots=g["rotation"])
        if self.track_stats:
            out["means2D"] = self.viewspace_grad
        self._packed_views = 0
        if self.factorize_sh or sh_via_colour:     # as in the event iteration, with one view: 3 floats per Gaussian are
            self._colour_gradients_instead_of_sh(out, settings)      # exchanged / kept instead of the 48 of the SH gradient
        rasterizer.backward_multi(raw, dpix, out, flags=self._backward_flags(raw, sh_via_colour),
                                  rank1={0: rasterizer.GRAY_WEIGHTS} if r1 else None)
        return loss, raw

    def step_image_autograd(self, cam, gt_image, bg, mode="gray", lambda_dssim=0.2):
        """The same iteration through torch autograd (torch activations, drop-in operator, autograd losses): the
        equivalence check of step_image."""
        self.sync_features()
        self.iteration += 1
        it = self.iteration
        self.flat_grad.zero_()
        leaves = {k: v.detach().requires_grad_(True) for k, v in self.views.items()}
        feats_ref = leaves["features"].t().reshape(self.N, 16, 3)
        scales, rots = torch.exp(leaves["scaling"]), torch.nn.functional.normalize(leaves["rotation"])
        opac = torch.sigmoid(leaves["opacity"])
        m2 = torch.zeros_like(leaves["xyz"], requires_grad=True)
        img, radii = rasterize_gaussians(leaves["xyz"], m2, feats_ref, None, opac, scales, rots, None,
                                         self._settings(cam, bg))
        fn = losses.gray_iteration_loss if mode == "gray" else losses.rgb_iteration_loss
        loss = fn(img, gt_image, lambda_dssim)
        loss.backward()
        for k, v in leaves.items():
            self.grads[k].copy_(v.grad)
        self.steps["gauss"] += 1; self.steps["opacity"] += 1
        self._adam(it, {"gauss": self.steps["gauss"], "opacity": self.steps["opacity"], "c": 0})
        return loss.detach()

    # ------------------------------------------------------------------ reference-style path (autograd)
    def step_autograd(self, cam_int, cam_now, cam_next, gt_int, gt_now, gt_next, bg, gt_blur=None):
        """Same iteration through torch autograd: torch activations (gaussian_model.py:95-118), the drop-in
        rasteriser operator with in-kernel SH, the autograd event loss.  Gradients land in flat_grad."""
        self.sync_features()
        self.iteration += 1
        it = self.iteration
        self.flat_grad.zero_()
        leaves = {k: v.detach().requires_grad_(True) for k, v in self.views.items()}
        feats_ref = leaves["features"].t().reshape(self.N, 16, 3)           # reference (N,16,3) layout for the op
        c = self.c.detach().clone().requires_grad_(True)
        scales, rots = torch.exp(leaves["scaling"]), torch.nn.functional.normalize(leaves["rotation"])
        opac = torch.sigmoid(leaves["opacity"])
        imgs = []
        for cam in (cam_int, cam_now, cam_next):
            m2 = torch.zeros_like(leaves["xyz"], requires_grad=True)
            img, radii = rasterize_gaussians(leaves["xyz"], m2, feats_ref, None, opac, scales, rots, None,
                                             self._settings(cam, bg))
            imgs.append(img)
        loss = losses.event_iteration_loss(imgs[0], imgs[1], imgs[2], c, gt_int, gt_now, gt_next, gt_blur)
        loss.backward()
        for k, v in leaves.items():
            self.grads[k].copy_(v.grad)
        self.c_grad.copy_(c.grad)
        for k in self.steps:
            self.steps[k] += 1
        self._adam(it, dict(self.steps))
        return loss
