"""distCUDA2: mean squared distance to the 3 nearest other points (scene/gaussian_model.py:134)."""
import torch

from . import _lib


def distCUDA2(points):
    if not points.is_cuda:
        raise RuntimeError("distCUDA2 expects a CUDA/HIP tensor (no CPU path)")
    pts = points.detach().to(torch.float32).contiguous()
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise RuntimeError("points must have dimensions (num_points, 3)")
    P = pts.shape[0]
    out = torch.zeros(P, dtype=torch.float32, device=pts.device)
    if P == 0:
        return out
    L = _lib.lib()
    scratch = torch.empty(L.e3dgs_knn_scratch_bytes(P), dtype=torch.uint8, device=pts.device)
    with torch.cuda.device(pts.device):
        rc = L.e3dgs_dist_knn3(P, _lib.ptr(pts), _lib.ptr(out), _lib.ptr(scratch), _lib.current_stream())
    _lib.check(rc, "e3dgs_dist_knn3")
    return out
