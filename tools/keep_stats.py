#!/usr/bin/env python
"""Lane utilisation of the compositing kernels on sampled tiles of the benchmark scene: of the (entry, 16x4 strip) pairs
the forward evaluates, how many hold a contributing pixel at all, and how many of the 64 pixels contribute?
(Vectorised torch re-composition of the sampled tiles with the kernels' rules; GPU box.)"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from event_3dgs_amd import synth, rasterizer
from event_3dgs_amd.cameras import orbit_camera
from event_3dgs_amd.train_step import EventTrainer
dev = torch.device("cuda", 0)
N, W, H = 1_000_000, 1920, 1080
params = synth.make_scene(N, "trained", seed=0, device=dev)
cam = orbit_camera(0, 64, W, H, device=dev)
tr = EventTrainer(params, dev)
raw = tr.render_raw(cam, torch.zeros(3, device=dev))
st = rasterizer.state_views(raw, N, W, H)
rg, pl = st["ranges"].long(), st["point_list"].long()
recA, recB, recC = st["recA"], st["recB"], st["recC"]
gx = (W + 15) // 16
tiles = torch.randperm(rg.shape[0], generator=torch.Generator().manual_seed(0))[:400].tolist()
ys, xs = torch.meshgrid(torch.arange(16, device=dev), torch.arange(16, device=dev), indexing="ij")
S = dict(entries=0, processed=0, eval_strips=0, contrib_strips=0, contrib_px=0, entries_any=0, dead_strip_tests=0, strip_tests=0,
         bwd_entries=0, bwd_strips=0)
for t in tiles:
    ids = pl[rg[t, 0]:rg[t, 1]]
    n = ids.numel()
    if n == 0:
        continue
    tx, ty = t % gx, t // gx
    px = (tx * 16 + xs).float()[None]; py = (ty * 16 + ys).float()[None]
    a, b, c = recA[ids], recB[ids], recC[ids]
    dx = a[:, 0, None, None] - px; dy = a[:, 1, None, None] - py
    power = -0.5 * (a[:, 2, None, None] * dx * dx + b[:, 0, None, None] * dy * dy) - a[:, 3, None, None] * dx * dy
    alpha = torch.clamp_max(b[:, 1, None, None] * torch.exp(power), 0.99)
    valid = (power <= 0) & (alpha >= 1.0 / 255.0) & (px < W) & (py < H)
    w = torch.where(valid, alpha, torch.zeros_like(alpha))
    Tafter = torch.cumprod(1 - w, dim=0)
    Tbefore = torch.cat((torch.ones_like(Tafter[:1]), Tafter[:-1]), 0)
    stop = valid & (Tafter < 1e-4)                      # this entry would push T below the threshold: pixel done, not applied
    done_before = torch.cat((torch.zeros_like(stop[:1]), (torch.cumsum(stop.int(), 0) > 0)[:-1]), 0)
    alive = ~done_before & (px < W) & (py < H)
    contrib = valid & alive & ~stop
    reach = (power >= c[:, 1, None, None])
    strips = lambda m: m.view(n, 4, 4, 16).any(dim=3).any(dim=2)          # (n, 4): rows 4k..4k+3
    tile_alive = alive.view(n, -1).any(dim=1)
    processed = int(tile_alive.sum())                  # entries the forward walks before every pixel is done (round granularity ignored)
    ev = strips(alive & reach) & tile_alive[:, None]
    cs = strips(contrib)
    S["entries"] += n; S["processed"] += processed
    S["eval_strips"] += int(ev.sum()); S["contrib_strips"] += int(cs.sum())
    # alternative 64-pixel groupings of the tile (round 6: would 8 x 8 blocks be evaluated less often than 16 x 4 strips?)
    blocks8 = lambda m: m.view(n, 2, 8, 2, 8).any(dim=4).any(dim=2).reshape(n, 4)
    cols4 = lambda m: m.view(n, 16, 4, 4).any(dim=3).any(dim=1)          # 4 x 16 column strips
    for name, part in (("b8", blocks8), ("c4", cols4)):
        S.setdefault("eval_" + name, 0); S.setdefault("contrib_" + name, 0)
        S["eval_" + name] += int((part(alive & reach) & tile_alive[:, None]).sum())
        S["contrib_" + name] += int(part(contrib).sum())
    S["contrib_px"] += int(contrib.sum()); S["entries_any"] += int(cs.any(dim=1).sum())
    alive_strip = strips(alive)
    S["strip_tests"] += 4 * processed; S["dead_strip_tests"] += int((~alive_strip & tile_alive[:, None]).sum())
    # backward: walks entries below the tile's largest n_contrib whose mask is non-zero; per strip: some pixel with
    # contributor <= last and power >= pmin
    last = (contrib * torch.arange(1, n + 1, device=dev)[:, None, None]).amax(dim=0)     # n_contrib per pixel
    walk = torch.arange(1, n + 1, device=dev) <= last.max()
    live_b = strips((torch.arange(1, n + 1, device=dev)[:, None, None] <= last[None]) & reach) & ev
    S["bwd_entries"] += int((walk & ev.any(dim=1)).sum()); S["bwd_strips"] += int((live_b & walk[:, None]).sum())
print(S)
print("forward: processed %.2f of the list; evaluated strips per processed entry %.2f; strips with a contributing pixel / evaluated %.2f; "
      "contributing pixels per evaluated strip %.1f of 64; strip tests on strips with no live pixel %.3f" % (
          S["processed"] / S["entries"], S["eval_strips"] / S["processed"], S["contrib_strips"] / S["eval_strips"],
          S["contrib_px"] / S["eval_strips"], S["dead_strip_tests"] / S["strip_tests"]))
print("groupings of 64 pixels, evaluated groups per processed entry (contributing groups): 16x4 strips %.3f (%.3f), 8x8 blocks "
      "%.3f (%.3f), 4x16 columns %.3f (%.3f)" % (S["eval_strips"] / S["processed"], S["contrib_strips"] / S["processed"],
                                                 S["eval_b8"] / S["processed"], S["contrib_b8"] / S["processed"],
                                                 S["eval_c4"] / S["processed"], S["contrib_c4"] / S["processed"]))
print("backward: visited entries %.2f of the list, live strips per visited entry %.2f, contributing pixels per live strip %.1f" % (
    S["bwd_entries"] / S["entries"], S["bwd_strips"] / max(S["bwd_entries"], 1), S["contrib_px"] / max(S["bwd_strips"], 1)))
