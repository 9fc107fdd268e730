"""ctypes binding of the C oracle (oracle/gs_oracle.c, oracle/aux_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the product packages.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _cpu_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    targets = [os.path.join(_HERE, n) for n in ("libgs_oracle.so", "libgs_oracle_fma.so")]
    srcs = [os.path.join(_HERE, n) for n in ("gs_oracle.c", "aux_oracle.c")]
    stale = force or any(
        not os.path.exists(t) or os.path.getmtime(t) < max(os.path.getmtime(s) for s in srcs) for t in targets
    )
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)
    return targets


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    name = "libgs_oracle_fma.so" if _cpu_has_fma() else "libgs_oracle.so"
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    fp, ip, vp = C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_void_p
    L.gso_forward.restype = vp
    L.gso_forward.argtypes = [C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int, fp, fp, fp, fp, fp, C.c_float, fp, fp,
                              fp, fp, fp, C.c_float, C.c_float, fp, ip]
    L.gso_num_rendered.restype = C.c_int64
    L.gso_num_rendered.argtypes = [vp]
    for n in ("depth", "xy", "conic_opacity", "rgb", "cov3d", "rect", "tiles_touched", "clamped", "keys",
              "point_list", "ranges", "final_T", "n_contrib"):
        f = getattr(L, "gso_" + n)
        f.restype = vp
        f.argtypes = [vp]
    L.gso_free.argtypes = [vp]
    L.gso_set_tile_rows.argtypes = [C.c_int, C.c_int]
    L.gso_timings.argtypes = [vp, C.POINTER(C.c_double)]
    L.gso_backward.argtypes = [vp, fp] + [fp] * 9
    L.gso_exp_det.restype = C.c_float
    L.gso_exp_det.argtypes = [C.c_float]
    L.gso_activate.argtypes = [C.c_int, fp, fp, fp, fp, fp, fp]
    L.gso_knn3.argtypes = [C.c_int, fp, fp]
    L.gso_event_loss.argtypes = [C.c_int, C.c_int] + [fp] * 7 + [C.c_float, C.c_float, fp, fp, fp,
                                                                 C.POINTER(C.c_double)]
    L.gso_adam.argtypes = [C.c_size_t, fp, fp, fp, fp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int]
    _LIB = L
    return L


def _f(a):
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _view(ptr, dtype, shape):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()


class Forward:
    """One forward pass of the C oracle; keeps the context for backward()."""

    def __init__(self, *, means3D, opacities, viewmatrix, projmatrix, campos, bg, width, height, tanfovx, tanfovy,
                 colors_precomp=None, shs=None, sh_degree=0, scales=None, rotations=None, cov3D_precomp=None,
                 scale_modifier=1.0, tile_rows=None):
        L = lib()
        if tile_rows is not None:
            L.gso_set_tile_rows(int(tile_rows[0]), int(tile_rows[1]))
        self._keep = []
        def f(a):
            arr, p = _f(a)
            self._keep.append(arr)
            return p
        means3D = np.ascontiguousarray(means3D, np.float32).reshape(-1, 3)
        P = means3D.shape[0]
        assert (shs is None) != (colors_precomp is None)
        assert (cov3D_precomp is None) != (scales is None or rotations is None)
        M = 0 if shs is None else np.asarray(shs).reshape(P, -1, 3).shape[1]
        self.P, self.W, self.H, self.M, self.D = P, int(width), int(height), M, int(sh_degree)
        self.gx, self.gy = (self.W + 15) // 16, (self.H + 15) // 16
        self.out_color = np.zeros((3, self.H, self.W), np.float32)
        self.radii = np.zeros(P, np.int32)
        self.has_scale_rot = cov3D_precomp is None
        self.has_sh = shs is not None
        self._ctx = L.gso_forward(
            P, self.D, M, f(bg), self.W, self.H, f(means3D), f(shs), f(colors_precomp),
            f(np.asarray(opacities).reshape(-1)), f(scales), float(scale_modifier), f(rotations), f(cov3D_precomp),
            f(np.asarray(viewmatrix).reshape(-1)), f(np.asarray(projmatrix).reshape(-1)), f(campos),
            float(tanfovx), float(tanfovy), self.out_color.ctypes.data_as(C.POINTER(C.c_float)),
            self.radii.ctypes.data_as(C.POINTER(C.c_int)))
        self.num_rendered = int(L.gso_num_rendered(self._ctx))
        tm = (C.c_double * 3)()
        L.gso_timings(self._ctx, tm)
        self.timings = tuple(tm)

    def _get(self, name, dtype, shape):
        return _view(getattr(lib(), "gso_" + name)(self._ctx), dtype, shape)

    @property
    def depth(self): return self._get("depth", np.float32, (self.P,))
    @property
    def xy(self): return self._get("xy", np.float32, (self.P, 2))
    @property
    def conic_opacity(self): return self._get("conic_opacity", np.float32, (self.P, 4))
    @property
    def rgb(self): return self._get("rgb", np.float32, (self.P, 3))
    @property
    def cov3d(self): return self._get("cov3d", np.float32, (self.P, 6))
    @property
    def rect(self): return self._get("rect", np.int32, (self.P, 4))
    @property
    def tiles_touched(self): return self._get("tiles_touched", np.uint32, (self.P,))
    @property
    def clamped(self): return self._get("clamped", np.uint8, (self.P, 3))
    @property
    def keys(self): return self._get("keys", np.uint64, (self.num_rendered,))
    @property
    def point_list(self): return self._get("point_list", np.uint32, (self.num_rendered,))
    @property
    def ranges(self): return self._get("ranges", np.uint32, (self.gx * self.gy, 2))
    @property
    def final_T(self): return self._get("final_T", np.float32, (self.H, self.W))
    @property
    def n_contrib(self): return self._get("n_contrib", np.uint32, (self.H, self.W))

    def backward(self, dL_dpix):
        """Returns a dict of gradients (numpy, fp32).  conic grads are TRUE partials (A,B,C)."""
        L = lib()
        P, M = self.P, self.M
        g = np.ascontiguousarray(dL_dpix, np.float32).reshape(3, self.H, self.W)
        out = {
            "means2D": np.zeros((P, 3), np.float32), "conic": np.zeros((P, 3), np.float32),
            "opacities": np.zeros((P,), np.float32), "colors": np.zeros((P, 3), np.float32),
            "means3D": np.zeros((P, 3), np.float32), "cov3D": np.zeros((P, 6), np.float32),
            "shs": np.zeros((P, max(M, 1), 3), np.float32) if self.has_sh else None,
            "scales": np.zeros((P, 3), np.float32) if self.has_scale_rot else None,
            "rotations": np.zeros((P, 4), np.float32) if self.has_scale_rot else None,
        }
        p = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))
        L.gso_backward(self._ctx, p(g), p(out["means2D"]), p(out["conic"]), p(out["opacities"]), p(out["colors"]),
                       p(out["means3D"]), p(out["cov3D"]), p(out["shs"]), p(out["scales"]), p(out["rotations"]))
        return out

    def close(self):
        if self._ctx:
            lib().gso_free(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def exp_det(x):
    L = lib()
    return np.array([L.gso_exp_det(float(v)) for v in np.asarray(x, np.float32).reshape(-1)], np.float32)


def activate(log_scales=None, raw_rotations=None, logit_opacities=None):
    """gso_activate(): the oracle's deterministic exp / normalize / sigmoid (scene/gaussian_model.py:33-41,95-118) of
    the RAW parameters -- what the E3DGS_FLAG_PREACT kernels compute.  Returns (scales, rotations, opacities)."""
    n = [np.asarray(a).reshape(-1, k).shape[0] for a, k in ((log_scales, 3), (raw_rotations, 4), (logit_opacities, 1))
         if a is not None]
    P = n[0]
    assert all(m == P for m in n)
    ins = [_f(None if a is None else np.asarray(a, np.float32).reshape(P, k))
           for a, k in ((log_scales, 3), (raw_rotations, 4), (logit_opacities, 1))]
    outs = [None if a is None else np.zeros((P, k), np.float32)
            for a, k in ((log_scales, 3), (raw_rotations, 4), (logit_opacities, 1))]
    p = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))
    lib().gso_activate(P, ins[0][1], ins[1][1], ins[2][1], p(outs[0]), p(outs[1]), p(outs[2]))
    return tuple(outs)


def activate_backward(raw_rotations, scales, rotations, opacities, g_scales, g_rotations, g_opacities):
    """Chain rule of activate() in float64: gradients w.r.t. the activated values -> w.r.t. the raw parameters
    (d exp = s, d sigmoid = o (1 - o), d normalize = (g - q^ (q^ . g)) / |q|)."""
    s, o = np.asarray(scales, np.float64), np.asarray(opacities, np.float64).reshape(-1, 1)
    qh = np.asarray(rotations, np.float64)
    nrm = np.maximum(np.linalg.norm(np.asarray(raw_rotations, np.float64), axis=1, keepdims=True), 1e-12)
    gs = np.asarray(g_scales, np.float64).reshape(s.shape) * s
    go = np.asarray(g_opacities, np.float64).reshape(o.shape) * o * (1.0 - o)
    gq = np.asarray(g_rotations, np.float64).reshape(qh.shape)
    gq = (gq - qh * (qh * gq).sum(axis=1, keepdims=True)) / nrm
    return gs, gq, go


def knn3(points):
    pts, pp = _f(np.asarray(points, np.float32).reshape(-1, 3))
    out = np.zeros(pts.shape[0], np.float32)
    lib().gso_knn3(pts.shape[0], pp, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def event_loss(image, now, nxt, gt_int, gt_now, gt_next, c, gt_c=0.17, gt_blur=None):
    H, W = np.asarray(image).shape[-2:]
    arrs = [_f(a) for a in (image, now, nxt, gt_int, gt_now, gt_next, gt_blur)]
    d_image, d_now, d_next = (np.zeros((3, H, W), np.float32) for _ in range(3))
    scal = np.zeros(8, np.float64)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    lib().gso_event_loss(W, H, *[a[1] for a in arrs], float(c), float(gt_c), fp(d_image), fp(d_now), fp(d_next),
                         scal.ctypes.data_as(C.POINTER(C.c_double)))
    return {"loss": scal[0], "dc": scal[1], "rho": scal[2], "l1_event": scal[3], "l1_int": scal[4],
            "l1_blur": scal[5], "d_image": d_image, "d_now": d_now, "d_next": d_next}


def adam(p, g, m, v, lr, b1, b2, eps, step):
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    for a in (p, g, m, v):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    lib().gso_adam(p.size, fp(p), fp(g), fp(m), fp(v), lr, b1, b2, eps, step)
