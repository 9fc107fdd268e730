#!/usr/bin/env python
"""How many 16x4 strips vs 8x8 blocks of a tile does an instance reach (alpha >= 1/255 possible)?  Sampled tiles of
the benchmark scene; informs the lane->pixel mapping of the compositing kernels."""
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
g = torch.Generator().manual_seed(0)
tiles = torch.randperm(rg.shape[0], generator=g)[:300].tolist()
tot = strips = blocks = 0
ys, xs = torch.meshgrid(torch.arange(16, device=dev), torch.arange(16, device=dev), indexing="ij")
for t in tiles:
    ids = pl[rg[t, 0]:rg[t, 1]]
    if ids.numel() == 0:
        continue
    tx, ty = t % gx, t // gx
    px = (tx * 16 + xs).float()[None]; py = (ty * 16 + ys).float()[None]
    a, b, c = recA[ids], recB[ids], recC[ids]
    dx = a[:, 0, None, None] - px; dy = a[:, 1, None, None] - py
    power = -0.5 * (a[:, 2, None, None] * dx * dx + b[:, 0, None, None] * dy * dy) - a[:, 3, None, None] * dx * dy
    live = power >= c[:, 1, None, None]                       # (n,16,16)
    s = live.view(-1, 4, 4, 16).any(dim=3).any(dim=2)          # strips: rows 4k..4k+3
    bl = live.view(-1, 2, 8, 2, 8).any(dim=4).any(dim=2).reshape(-1, 4)
    tot += ids.numel(); strips += int(s.sum()); blocks += int(bl.sum())
print("instances", tot, "live 16x4 strips / instance", strips / tot, "live 8x8 blocks / instance", blocks / tot)
