"""Event iteration loss, fused on the GPU through the C ABI.

event_iteration_loss mirrors the loss block of train.py:165-203 built from
utils/loss_utils.py:differentialable_event_simu (:234-249), rgb_to_LUVscale (:24-28) and
l1_loss (:270-271); lambda_dssim is forced to 0 there (train.py:177) so the SSIM term vanishes.
"""
import torch

from . import _lib


class _EventLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, img_now, img_next, c, gt_int, gt_now, gt_next, gt_blur, gt_c):
        L = _lib.lib()
        dev = image.device
        _, H, W = image.shape
        prep = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        image_c, now_c, next_c = prep(image), prep(img_now), prep(img_next)
        if now_c.data_ptr() == image_c.data_ptr():
            # (the kernel reads "same input, separate outputs" as the shared-render convention of include/e3dgs_hip.h:
            # d_now = total.  Autograd adds the two gradients of one tensor itself, so it gets two distinct inputs.)
            now_c = now_c.clone()
        gi, gn, gx, gb = prep(gt_int), prep(gt_now), prep(gt_next), prep(gt_blur)
        c_dev = c.detach().to(torch.float32).reshape(1).contiguous()
        d_image, d_now, d_next = (torch.empty_like(image_c) for _ in range(3))
        scalars = torch.empty(8, dtype=torch.float32, device=dev)
        scratch = torch.empty(L.e3dgs_event_loss_scratch_bytes(W, H), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = L.e3dgs_event_loss(W, H, _lib.ptr(image_c), _lib.ptr(now_c), _lib.ptr(next_c), _lib.ptr(gi),
                                    _lib.ptr(gn), _lib.ptr(gx), _lib.ptr(gb), _lib.ptr(c_dev), float(gt_c),
                                    _lib.ptr(d_image), _lib.ptr(d_now), _lib.ptr(d_next), _lib.ptr(scalars), None,
                                    _lib.ptr(scratch), _lib.current_stream())
        _lib.check(rc, "e3dgs_event_loss")
        ctx.save_for_backward(d_image, d_now, d_next, scalars)
        ctx.c_shape = c.shape
        ctx.stats = scalars
        return scalars[0].clone()

    @staticmethod
    def backward(ctx, g):
        d_image, d_now, d_next, scalars = ctx.saved_tensors
        return (d_image * g, d_now * g, d_next * g, (scalars[1] * g).reshape(ctx.c_shape), None, None, None, None,
                None)


class PairCounts:
    """count(D* != 0) of the ground-truth pairs an event loop has met (e3dgs_event_loss_cached): one device double per
    (gt_now, gt_next) pair, written by the first iteration that sees the pair and handed back afterwards, so that the loss
    sweeps the images once instead of twice.  Entries are matched by tensor IDENTITY (weak references: an address can be
    reused by another tensor) and in-place version; at most `capacity` pairs are remembered (oldest dropped).  All calls
    that share an instance must be enqueued on one stream (the count is produced and consumed in stream order)."""

    def __init__(self, capacity=4096):
        self.capacity = int(capacity)
        self._entries = {}

    @staticmethod
    def _sig(gt_now, gt_next, gt_c):
        # (versions catch in-place torch ops; the data pointers catch set_() / .data swaps.  A write through a raw pointer or
        # DLPack that bumps neither is invisible here: such a caller passes pair_counts=None or starts a fresh PairCounts)
        return (gt_now._version, gt_next._version, gt_now.data_ptr(), gt_next.data_ptr(), float(gt_c), tuple(gt_now.shape))

    def lookup(self, gt_now, gt_next, gt_c):
        """The pair's count tensor, or None when the pair is new (or was modified in place since)."""
        key = (id(gt_now), id(gt_next))
        e = self._entries.get(key)
        if e is None:
            return None
        if e[0]() is gt_now and e[1]() is gt_next and e[2] == self._sig(gt_now, gt_next, gt_c):
            return e[3]
        if e[0]() is None or e[1]() is None:            # the frames are gone (their ids may be reused): drop the entry
            del self._entries[key]
        return None

    def purge(self):
        """Drop the entries whose frames have been freed (called when the table fills up)."""
        for key in [k for k, e in self._entries.items() if e[0]() is None or e[1]() is None]:
            del self._entries[key]

    def remember(self, gt_now, gt_next, gt_c, count):
        """`count`: the device double an ENQUEUED e3dgs_event_loss_cached(nz_valid = 0) call fills."""
        import weakref
        key = (id(gt_now), id(gt_next))
        self._entries.pop(key, None)
        if len(self._entries) >= self.capacity:
            self.purge()
        while len(self._entries) >= self.capacity:
            self._entries.pop(next(iter(self._entries)))
        self._entries[key] = (weakref.ref(gt_now), weakref.ref(gt_next), self._sig(gt_now, gt_next, gt_c), count)


def event_loss_raw(image, img_now, img_next, c, gt_int, gt_now, gt_next, gt_blur=None, gt_c=0.17, out=None, dc_out=None,
                   pair_counts=None, rank1=False):
    """Direct call of e3dgs_event_loss (no autograd; pair_counts: a PairCounts instance -> e3dgs_event_loss_cached, one
    sweep over the images from the second time a ground-truth pair is met, bit-identical results).  Returns (scalars[8], d_image, d_now, d_next):
    scalars[0] = loss, [1] = dL/dc, [2] = rho, [3..5] = L1 event / intensity / blur.  `out` may carry
    preallocated (scalars, d_image, d_now, d_next, scratch) tensors to reuse across steps; `dc_out`: a one-element
    device tensor that also receives dL/dc (the threshold's slot of a flat gradient buffer).
    Shared render (image IS img_now): with d_now IS d_image the sum of the two gradients is stored; with separate
    outputs d_now receives the sum and d_image the intensity term's part alone (include/e3dgs_hip.h).
    rank1: e3dgs_event_loss_rank1 -- d_next, and d_now when it is a render of its own, receive in plane 0 the scalar field
    s with dL/dC = s * (0.4124, 0.35758, 0.1804) (rasterizer.LUV_WEIGHTS; the input of backward_multi(rank1=...)); their
    planes 1 and 2 are left as they are."""
    L = _lib.lib()
    dev = image.device
    _, H, W = image.shape
    for t in (image, img_now, img_next, gt_int, gt_now, gt_next):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("event_loss_raw needs contiguous fp32 GPU tensors")
    if out is None:
        out = (torch.empty(8, dtype=torch.float32, device=dev), torch.empty_like(image), torch.empty_like(image),
               torch.empty_like(image),
               torch.empty(L.e3dgs_event_loss_scratch_bytes(W, H), dtype=torch.uint8, device=dev))
    scalars, d_image, d_now, d_next, scratch = out
    head = (W, H, _lib.ptr(image), _lib.ptr(img_now), _lib.ptr(img_next), _lib.ptr(gt_int), _lib.ptr(gt_now),
            _lib.ptr(gt_next), _lib.ptr(gt_blur), _lib.ptr(c), float(gt_c), _lib.ptr(d_image), _lib.ptr(d_now),
            _lib.ptr(d_next), _lib.ptr(scalars), _lib.ptr(dc_out))
    with torch.cuda.device(dev):
        if pair_counts is None:
            if rank1:
                rc = L.e3dgs_event_loss_rank1(*head, None, 0, _lib.ptr(scratch), _lib.current_stream())
            else:
                rc = L.e3dgs_event_loss(*head, _lib.ptr(scratch), _lib.current_stream())
        else:
            cnt = pair_counts.lookup(gt_now, gt_next, gt_c)
            valid = cnt is not None
            if not valid:
                cnt = torch.zeros(1, dtype=torch.float64, device=dev)
            fn = L.e3dgs_event_loss_rank1 if rank1 else L.e3dgs_event_loss_cached
            rc = fn(*head, _lib.ptr(cnt), int(valid), _lib.ptr(scratch), _lib.current_stream())
            if rc == 0 and not valid:
                pair_counts.remember(gt_now, gt_next, gt_c, cnt)
    _lib.check(rc, "e3dgs_event_loss")
    return scalars, d_image, d_now, d_next


def event_iteration_loss(image, img_now, img_next, c, gt_int, gt_now, gt_next, gt_blur=None, gt_c=0.17):
    """loss of train.py:165-203; `c` is the learnable contrast threshold (train.py:71-73)."""
    if not image.is_cuda:
        raise RuntimeError("event_iteration_loss runs on the GPU only (no CPU path)")
    return _EventLoss.apply(image, img_now, img_next, c, gt_int, gt_now, gt_next, gt_blur, gt_c)


def adam_step_(param, grad, exp_avg, exp_avg_sq, lr, step, beta1=0.9, beta2=0.999, eps=1e-15, lr_b=0.0, period=0,
               split=0):
    """In-place fused Adam on one flat fp32 tensor (train.py:330-332; eps of gaussian_model.py:163)."""
    for t in (param, grad, exp_avg, exp_avg_sq):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("adam_step_ needs contiguous fp32 GPU tensors")
    with torch.cuda.device(param.device):
        rc = _lib.lib().e3dgs_adam_step(param.numel(), _lib.ptr(param), _lib.ptr(grad), _lib.ptr(exp_avg),
                                        _lib.ptr(exp_avg_sq), float(lr), beta1, beta2, eps, int(step), float(lr_b),
                                        int(period), int(split), _lib.current_stream())
    _lib.check(rc, "e3dgs_adam_step")
    _bump_versions(param, exp_avg, exp_avg_sq)


def _bump_versions(*tensors):
    """The kernels above write through raw pointers, which autograd's version counters do not see: bump them, so that
    everything keyed on `tensor._version` (adopt._features' coefficient cache, autograd's saved-tensor check, PairCounts)
    notices the update as it would notice an in-place torch op."""
    for t in tensors:
        torch.autograd.graph.increment_version(t)


def adam_step_segments_(param, grad, exp_avg, exp_avg_sq, seg_end, lrs, eps, step, beta1=0.9, beta2=0.999, gap=None):
    """In-place fused Adam over consecutive segments of one flat fp32 tensor, each with its own learning rate and eps, in
    one launch.  seg_end: ascending element offsets, the last one == param.numel().  `step`: one int for all segments
    (e3dgs_adam_step_segments) or one per segment, <= 0 = leave that segment untouched (e3dgs_adam_step_groups: torch's
    per-parameter step counts and its skipping of parameters without a gradient).  gap = (begin, length): elements the
    launch does not visit at all (e3dgs_adam_step_groups_gap; needs one step count per segment)."""
    import ctypes as C
    for t in (param, grad, exp_avg, exp_avg_sq):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("adam_step_segments_ needs contiguous fp32 GPU tensors")
    k = len(seg_end)
    common = (param.numel(), _lib.ptr(param), _lib.ptr(grad), _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq), k,
              (C.c_size_t * k)(*[int(e) for e in seg_end]), (C.c_float * k)(*[float(x) for x in lrs]),
              (C.c_float * k)(*[float(x) for x in eps]), beta1, beta2)
    with torch.cuda.device(param.device):
        if isinstance(step, (tuple, list)):
            if len(step) != k:
                raise ValueError("one step count per segment")
            steps = (C.c_int * k)(*[int(x) for x in step])
            if gap is not None:
                rc = _lib.lib().e3dgs_adam_step_groups_gap(*common, steps, int(gap[0]), int(gap[1]), _lib.current_stream())
            else:
                rc = _lib.lib().e3dgs_adam_step_groups(*common, steps, _lib.current_stream())
        elif gap is not None:
            raise ValueError("a gap needs one step count per segment")
        else:
            rc = _lib.lib().e3dgs_adam_step_segments(*common, int(step), _lib.current_stream())
    _lib.check(rc, "e3dgs_adam_step_segments")
    _bump_versions(param, exp_avg, exp_avg_sq)


# ---------------------------------------------------------------------------------------------------------------
# SSIM / gray losses / PSNR: utils/loss_utils.py:18-23,40-48,359-418; utils/image_utils.py:19-21
class _SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2, to_gray):
        L = _lib.lib()
        if not img1.is_cuda:
            raise RuntimeError("ssim runs on the GPU only (no CPU path)")
        a = img1.detach().to(torch.float32).contiguous()
        b = img2.detach().to(torch.float32).contiguous()
        C, H, W = a.shape[-3:]
        out = torch.empty(1, dtype=torch.float32, device=a.device)
        need = img1.requires_grad
        d = torch.empty_like(a) if need else None
        scratch = torch.empty(L.e3dgs_ssim_scratch_bytes(C, H, W), dtype=torch.uint8, device=a.device)
        with torch.cuda.device(a.device):
            rc = L.e3dgs_ssim(C, H, W, int(bool(to_gray)), _lib.ptr(a), _lib.ptr(b), _lib.ptr(out), _lib.ptr(d),
                              _lib.ptr(scratch), _lib.current_stream())
        _lib.check(rc, "e3dgs_ssim")
        ctx.save_for_backward(d)
        ctx.shape = img1.shape
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return (d * g).reshape(ctx.shape) if d is not None else None, None, None


def image_loss_raw(image, gt_image, gray, lambda_dssim=0.2, out=None, rank1=False):
    """e3dgs_image_loss: (1 - lambda) L1 + lambda (1 - SSIM) of the --gray (train.py:213-223) or RGB (train.py:292-296)
    iteration and its gradient w.r.t. `image`, three launches, no autograd.  Returns (scalars[4], d_image): scalars[0] =
    loss, [1] = L1, [2] = SSIM.  `out` may carry preallocated (scalars, d_image, scratch).
    rank1 (gray only): e3dgs_image_loss_rank1 -- plane 0 of d_image receives the scalar field s with
    d loss / d image = s * (0.299, 0.587, 0.114) (rasterizer.GRAY_WEIGHTS); planes 1 and 2 are left as they are."""
    L = _lib.lib()
    for t in (image, gt_image):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("image_loss_raw needs contiguous fp32 GPU tensors")
    C_, H, W = image.shape
    if out is None:
        out = (torch.empty(4, dtype=torch.float32, device=image.device), torch.empty_like(image),
               torch.empty(L.e3dgs_image_loss_scratch_bytes(C_, H, W), dtype=torch.uint8, device=image.device))
    scalars, d_image, scratch = out
    if rank1 and not (gray and C_ == 3):
        raise ValueError("a rank-1 image gradient needs the gray loss of a 3-channel render")
    with torch.cuda.device(image.device):
        if rank1:
            rc = L.e3dgs_image_loss_rank1(H, W, float(lambda_dssim), _lib.ptr(image), _lib.ptr(gt_image),
                                          _lib.ptr(scalars), _lib.ptr(d_image), _lib.ptr(scratch), _lib.current_stream())
        else:
            rc = L.e3dgs_image_loss(C_, H, W, int(bool(gray)), float(lambda_dssim), _lib.ptr(image), _lib.ptr(gt_image),
                                    _lib.ptr(scalars), _lib.ptr(d_image), _lib.ptr(scratch), _lib.current_stream())
    _lib.check(rc, "e3dgs_image_loss")
    return scalars, d_image


def ssim(img1, img2):
    """utils/loss_utils.py:388-396 (size_average=True)."""
    return _SSIM.apply(img1, img2, False)


def ssim_gray(img1, img2):
    """utils/loss_utils.py:368-385: 3-channel inputs are converted with rgb_to_grayscale first."""
    if img1.shape[-3] == 3 and img2.shape[-3] == 3:
        return _SSIM.apply(img1, img2, True)
    return _SSIM.apply(img1, img2, False)


def rgb_to_grayscale(image):
    """utils/loss_utils.py:18-23"""
    return (0.299 * image[0] + 0.587 * image[1] + 0.114 * image[2]).unsqueeze(0)


def l1_loss_gray(network_output, gt):
    """utils/loss_utils.py:40-48"""
    if network_output.shape[-3] == 3 and gt.shape[-3] == 3:
        return torch.abs(rgb_to_grayscale(network_output) - rgb_to_grayscale(gt)).mean()
    return torch.abs(network_output - gt).mean()


def l1_loss(network_output, gt):
    """utils/loss_utils.py:270-271"""
    return torch.abs(network_output - gt).mean()


def gray_iteration_loss(image, gt_image, lambda_dssim=0.2):
    """The --gray (no event) loss of train.py:213-223."""
    return (1.0 - lambda_dssim) * l1_loss_gray(image, gt_image) + lambda_dssim * (1.0 - ssim_gray(image, gt_image))


def rgb_iteration_loss(image, gt_image, lambda_dssim=0.2):
    """The RGB loss of train.py:292-296."""
    return (1.0 - lambda_dssim) * l1_loss(image, gt_image) + lambda_dssim * (1.0 - ssim(image, gt_image))


def psnr(img1, img2):
    """utils/image_utils.py:19-21"""
    mse = ((img1 - img2) ** 2).reshape(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))
