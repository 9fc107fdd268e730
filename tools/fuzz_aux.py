"""Randomised differential test of the small kernels on the per-iteration path against the oracle, on odd shapes:
distCUDA2 (point counts from 1, duplicates, clusters, collinear points), the event loss (frames from 1x1, zeros in
the targets, deblur term on/off), SSIM value + gradient (frames smaller than the 11x11 window) and Adam.

Usage (GPU box, repo root):  python tools/fuzz_aux.py [cases] [first_seed]
TEST INFRASTRUCTURE: imports oracle/ (never the product does).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from event_3dgs_amd import losses                   # noqa: E402
from oracle import c_oracle, torch_oracle            # noqa: E402

DEV = "cuda:0"


def check_knn(seed):
    from simple_knn._C import distCUDA2
    r = np.random.default_rng(seed)
    P = int(r.choice([1, 2, 3, 4, 5, 17, 64, 65, 300, 1023, 1025, 3000]))
    pts = (r.random((P, 3)) * r.choice([1e-3, 1.0, 50.0]) + r.choice([0.0, -20.0, 1000.0])).astype(np.float32)
    kind = r.integers(0, 5)
    if kind == 1 and P > 4:
        pts[P // 2:] = pts[: P - P // 2]                   # every point duplicated
    elif kind == 2:
        pts[:, 1:] = 0.0                                   # collinear
    elif kind == 3 and P > 10:
        pts[: P // 3] = pts[0] + (pts[: P // 3] - pts[0]) * 1e-4        # tight cluster + sparse rest
    elif kind == 4:
        pts[:] = pts[0]                                    # all identical
    got = distCUDA2(torch.from_numpy(pts).to(DEV)).cpu().numpy()
    ref = c_oracle.knn3(pts)
    ok = np.allclose(got, ref, rtol=3e-5, atol=1e-12, equal_nan=True) and got.shape == (P,)
    return "knn seed %d P=%d kind=%d" % (seed, P, kind), ([] if ok else ["max rel diff %.3g" % float(
        np.nanmax(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-20)))])


def check_event(seed):
    r = np.random.default_rng(seed)
    H, W = int(r.choice([1, 2, 15, 16, 17, 40, 97])), int(r.choice([1, 3, 16, 33, 64, 130]))
    q8 = lambda a: np.round(np.clip(a, 0, 1) * 255) / 255
    img = [r.random((3, H, W)).astype(np.float32) for _ in range(3)]
    gts = [q8(r.random((3, H, W))).astype(np.float32) for _ in range(3)]
    if r.random() < 0.5:
        gts[2] = gts[1].copy()                             # contrast target exactly zero everywhere
    if r.random() < 0.3:
        gts[1][:, : H // 2] = 0.0                          # black: ln(1e-8)
    blur = q8(r.random((3, H, W))).astype(np.float32) if r.random() < 0.4 else None
    c = float(r.choice([0.17, 0.05, 0.6]))
    t = lambda a: None if a is None else torch.from_numpy(a).to(DEV)
    cc = torch.tensor([c], device=DEV)
    scal, d0, d1, d2 = losses.event_loss_raw(t(img[0]), t(img[1]), t(img[2]), cc, t(gts[0]), t(gts[1]), t(gts[2]),
                                             gt_blur=t(blur))[:4]
    ref = c_oracle.event_loss(img[0], img[1], img[2], gts[0], gts[1], gts[2], c, gt_blur=blur)
    s = scal.cpu().numpy()
    problems = []
    # the one-sweep form (e3dgs_event_loss_cached) on the same pair: first call fills the pair's count, second call uses
    # it; both must equal the three-launch form bit for bit
    pc = losses.PairCounts()
    ti, tg = [t(a) for a in img], [t(a) for a in gts]
    tb = t(blur)
    for rep in range(2):
        got = losses.event_loss_raw(ti[0], ti[1], ti[2], cc, tg[0], tg[1], tg[2], gt_blur=tb, pair_counts=pc)
        for name, a, b in zip(("scalars", "d_image", "d_now", "d_next"), got, (scal, d0, d1, d2)):
            if not torch.equal(a, b):
                problems.append("cached call %d: %s differs from the three-launch form" % (rep, name))
    rel = lambda a, b: abs(a - b) / max(abs(b), 1e-6)
    if rel(float(s[0]), ref["loss"]) > 2e-5:
        problems.append("loss %.7g vs %.7g" % (s[0], ref["loss"]))
    if rel(float(s[1]), ref["dc"]) > 1e-4:
        problems.append("dc %.7g vs %.7g" % (s[1], ref["dc"]))
    for name, a, b in (("d_image", d0, ref["d_image"]), ("d_now", d1, ref["d_now"]), ("d_next", d2, ref["d_next"])):
        a = a.cpu().numpy()
        scale = max(float(np.abs(b).max()), 1e-12)
        if not np.isfinite(a).all() or float(np.abs(a - b).max()) > 1e-5 * scale + 1e-12:
            problems.append("%s max diff %.3g (scale %.3g)" % (name, float(np.abs(a - b).max()), scale))
    return "event seed %d %dx%d blur=%s c=%.2f" % (seed, W, H, blur is not None, c), problems


def check_ssim(seed):
    r = np.random.default_rng(seed)
    H, W = int(r.choice([1, 5, 10, 11, 12, 31, 64, 75])), int(r.choice([1, 4, 11, 16, 47, 128]))
    C = int(r.choice([1, 3]))
    a = torch.from_numpy(r.random((C, H, W)).astype(np.float32))
    b = torch.from_numpy(np.clip(a.numpy() + r.normal(0, 0.1, (C, H, W)), 0, 1).astype(np.float32))
    x = a.clone().to(DEV).requires_grad_(True)
    val = losses.ssim(x, b.to(DEV))
    val.backward()
    xr = a.clone().double().requires_grad_(True)
    ref = torch_oracle.ssim(xr, b.double())
    ref.backward()
    problems = []
    if abs(float(val.detach()) - float(ref.detach())) > 2e-5:
        problems.append("value %.7g vs %.7g" % (float(val.detach()), float(ref.detach())))
    g, gr = x.grad.cpu().numpy(), xr.grad.numpy()
    if float(np.abs(g - gr).max()) > 2e-4 * max(float(np.abs(gr).max()), 1e-9):
        problems.append("grad max diff %.3g (scale %.3g)" % (float(np.abs(g - gr).max()), float(np.abs(gr).max())))
    return "ssim seed %d C=%d %dx%d" % (seed, C, W, H), problems


def check_image_loss(seed):
    """e3dgs_image_loss (fused L1 + SSIM loss of the --gray / RGB iterations) against the fp64 restatement."""
    r = np.random.default_rng(seed)
    H, W = int(r.choice([1, 7, 16, 17, 40, 75])), int(r.choice([1, 5, 16, 33, 130]))
    gray = bool(r.random() < 0.5)
    lam = float(r.choice([0.2, 0.0, 1.0, 0.7]))
    a = torch.from_numpy(r.random((3, H, W)).astype(np.float32))
    b = torch.from_numpy(np.clip(a.numpy() + r.normal(0, 0.1, (3, H, W)), 0, 1).astype(np.float32))
    sc, d = losses.image_loss_raw(a.to(DEV), b.to(DEV), gray, lam)
    x = a.clone().double().requires_grad_(True)
    y = b.double()
    if gray:
        ref = torch_oracle.gray_iteration_loss(x, y, lam)
    else:
        ref = (1.0 - lam) * (x - y).abs().mean() + lam * (1.0 - torch_oracle.ssim(x, y))
    ref.backward()
    problems = []
    if abs(float(sc[0]) - float(ref.detach())) > 3e-5:
        problems.append("loss %.7g vs %.7g" % (float(sc[0]), float(ref.detach())))
    g, gr = d.cpu().numpy(), x.grad.numpy()
    # sign(img - gt) flips where the difference is ~0 in fp32: compare away from those pixels
    e = (a - b).numpy() if not gray else np.tile((0.299 * (a - b)[0] + 0.587 * (a - b)[1] + 0.114 * (a - b)[2]).numpy()[None], (3, 1, 1))
    ok = np.abs(e) > 1e-6
    if ok.any() and float(np.abs(g - gr)[ok].max()) > 2e-4 * max(float(np.abs(gr).max()), 1e-9):
        problems.append("grad max diff %.3g (scale %.3g)" % (float(np.abs(g - gr)[ok].max()), float(np.abs(gr).max())))
    return "image_loss seed %d %dx%d gray=%s lambda=%.1f" % (seed, W, H, gray, lam), problems


def check_adam(seed):
    r = np.random.default_rng(seed)
    n = int(r.choice([1, 3, 255, 256, 257, 4099, 100_003]))
    p, g = r.normal(0, 1, n).astype(np.float32), (r.normal(0, 1, n) * r.choice([1e-8, 1.0, 1e4])).astype(np.float32)
    m, v = np.zeros(n, np.float32), np.zeros(n, np.float32)
    if r.random() < 0.3:
        g[::2] = 0.0
    tp, tg, tm, tv = (torch.from_numpy(a.copy()).to(DEV) for a in (p, g, m, v))
    lr = float(r.choice([1.6e-4, 0.05, 0.0025]))
    for step in range(1, 4):
        losses.adam_step_(tp, tg, tm, tv, lr, step)
        c_oracle.adam(p, g, m, v, lr, 0.9, 0.999, 1e-15, step)
    problems = []
    for name, a, b in (("p", tp, p), ("m", tm, m), ("v", tv, v)):
        a = a.cpu().numpy()
        if not np.allclose(a, b, rtol=3e-6, atol=1e-30):
            problems.append("%s max rel %.3g" % (name, float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)))))
    return "adam seed %d n=%d lr=%g" % (seed, n, lr), problems


CHECKS = (check_knn, check_event, check_ssim, check_image_loss, check_adam)

if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for seed in range(first, first + cases):
        for fn in CHECKS:
            if os.environ.get("FUZZ_VERBOSE") == "1":
                print(fn.__name__, seed, flush=True)
            desc, problems = fn(seed)
            if problems:
                bad += 1
                print("FAIL", desc, "|", "; ".join(problems), flush=True)
    print("fuzz_aux: %d seeds x %d kernels, %d failing" % (cases, len(CHECKS), bad))
    sys.exit(1 if bad else 0)
