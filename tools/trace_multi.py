#!/usr/bin/env python
"""Occupancy profile over time of the 3-view compositing launches (per-tile wall_clock stamps written by the kernels
when e3dgs_debug_set_trace is armed): how long is the tail of the LPT-ordered launch?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from event_3dgs_amd import _lib, synth, rasterizer
from event_3dgs_amd.cameras import orbit_camera
from event_3dgs_amd.train_step import EventTrainer
dev = torch.device("cuda:0")
L = _lib.lib()
L.e3dgs_debug_set_trace.argtypes = [C.c_void_p]
N, W, H = (int(a) for a in (sys.argv[1:4] if len(sys.argv) >= 4 else (1_000_000, 1920, 1080)))
tr = EventTrainer(synth.make_scene(N, "trained", seed=0, device=dev), dev)
cams = [orbit_camera(0, 64, W, H, device=dev, daz=d) for d in (0.0, 0.005, 0.015)]
bg = torch.zeros(3, device=dev)
v = tr.views
T = 3 * ((W + 15) // 16) * ((H + 15) // 16)
settings = [tr._settings(c, bg) for c in cams]
names = dict(means3D=v["xyz"], sh=v["features"], opacities=v["opacity"], scales=v["scaling"], rots=v["rotation"])
dpix = torch.randn(3, 3, H, W, device=dev)

def profile(which):
    buf = torch.zeros(T * 6, dtype=torch.int64, device=dev)
    for rep in range(3):
        arm_f = buf.data_ptr() if (rep == 2 and which == "fwd") else None
        L.e3dgs_debug_set_trace(arm_f)
        raw = rasterizer.forward_multi(v["xyz"], v["features"], v["opacity"], v["scaling"], v["rotation"], settings,
                                       flags=tr.FWD_FLAGS)
        torch.cuda.synchronize()
        L.e3dgs_debug_set_trace(buf.data_ptr() if (rep == 2 and which == "bwd") else None)
        out = {n: torch.empty_like(t) for n, t in names.items()}
        rasterizer.backward_multi(raw, dpix, out)
        torch.cuda.synchronize()
        L.e3dgs_debug_set_trace(None)
    t = buf.cpu().numpy().reshape(T, 6)
    t = t[t[:, 1] > 0]
    T0, span = t[:, 0].min(), t[:, 1].max() - t[:, 0].min()
    edges = np.linspace(0, span, 21)
    act = [int(((t[:, 0] - T0) < edges[i + 1]).sum() - ((t[:, 1] - T0) < edges[i]).sum()) for i in range(20)]
    dur = t[:, 1] - t[:, 0]
    print(which, "tiles", len(t), "span(10ns)", span, "mean active waves", round(dur.sum() / span), "of 5120/8192 slots; profile", act)
    n = (t[:, 2] & 0xFFFFFFFF).astype(np.float64)        # walked entries of the tile
    o = np.argsort(-dur)
    print("   longest tiles: duration / span", np.round(dur[o[:5]] / span, 3), "entries", n[o[:5]].astype(int),
          "start / span", np.round((t[o[:5], 0] - T0) / span, 3))
    print("   entries per tile: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f;  sum %.3g" % (
        n.mean(), *np.percentile(n, [50, 90, 99]), n.max(), n.sum()))
    ok = n > 64
    per = dur[ok] * 10.0 / n[ok]                          # ns per walked entry
    for lo, hi in ((0.0, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.0)):
        m = ((t[ok, 0] - T0) >= lo * span) & ((t[ok, 0] - T0) < hi * span)
        if m.any(): print("   tiles starting in [%.2f, %.2f) of the span: %5d, ns per walked entry %.1f" % (lo, hi, m.sum(), per[m].mean()))
    hw = t[:, 3] & 0xFFFFFFFF; xcc = (t[:, 3] >> 32) & 0xF
    simd = (xcc << 16) | (hw & 0xFF30)                    # XCC | SE 15:13, SH 12, CU 11:8, SIMD 5:4 (HW_ID)
    ids, inv = np.unique(simd, return_inverse=True)
    fin = np.zeros(len(ids)); np.maximum.at(fin, inv, (t[:, 1] - T0).astype(np.float64))
    ent = np.zeros(len(ids)); np.add.at(ent, inv, n)
    cnt = np.bincount(inv)
    print("   SIMDs seen %d; tiles per SIMD min %d max %d; finishing time / span: p5 %.2f p25 %.2f p50 %.2f p75 %.2f p95 %.2f; "
          "entries per SIMD mean %.0f min %.0f max %.0f; corr(entries, finish) %.2f" % (
              len(ids), cnt.min(), cnt.max(), *np.percentile(fin / span, [5, 25, 50, 75, 95]), ent.mean(), ent.min(), ent.max(),
              np.corrcoef(ent, fin)[0, 1]))
    cu = simd >> 6
    idc, invc = np.unique(cu, return_inverse=True)
    finc = np.zeros(len(idc)); np.maximum.at(finc, invc, (t[:, 1] - T0).astype(np.float64))
    entc = np.zeros(len(idc)); np.add.at(entc, invc, n)
    print("   CUs seen %d; finishing time / span p5 %.2f p50 %.2f p95 %.2f; entries per CU min %.0f mean %.0f max %.0f; corr %.2f" % (
        len(idc), *np.percentile(finc / span, [5, 50, 95]), entc.min(), entc.mean(), entc.max(), np.corrcoef(entc, finc)[0, 1]))
    # which workgroup (launch slot / 4) went where: the dispatcher's pattern for the first 16 workgroups
    late = (t[:, 1] - T0) > 0.8 * span
    print("   tiles still running after 80 %% of the span: %d" % late.sum())

profile("fwd")
profile("bwd")
