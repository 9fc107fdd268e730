#!/usr/bin/env python
"""Statistics of the forward's per-entry strip masks over the entries the backward walks (GPU box)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from event_3dgs_amd import _lib, rasterizer, synth
from event_3dgs_amd.cameras import orbit_camera
from event_3dgs_amd.train_step import EventTrainer

N, W, H = 1_000_000, 1920, 1080
dev = "cuda:0"
tr = EventTrainer(synth.make_scene(N, "trained", seed=0, device=dev), dev)
cam = orbit_camera(0, 64, W, H, device=dev)
raw = tr.render_raw(cam, torch.zeros(3, device=dev))
I = raw["num_rendered"]
st = rasterizer.state_views(raw, N, W, H)
off = _lib.lib().e3dgs_state_offset_emit_gid(I) + (4 * I + 255) // 256 * 256
mask = raw["binning"][off:off + I].to(torch.int64)
rg = st["ranges"].to(torch.int64)
gx, gy = (W + 15) // 16, (H + 15) // 16
nc = torch.zeros(gy * 16, gx * 16, dtype=torch.int64, device=dev)
nc[:H, :W] = st["n_contrib"].to(torch.int64)
maxc = nc.reshape(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(gy * gx, 256).max(dim=1).values
pos = torch.arange(I, device=dev)
tile = torch.searchsorted(rg[:, 1].contiguous(), pos, right=True)
walked = (pos - rg[tile, 0]) < maxc[tile]
m = mask[walked]
pc = (m & 1) + ((m >> 1) & 1) + ((m >> 2) & 1) + ((m >> 3) & 1)
print("instances", I, "walked by backward", int(walked.sum()), "list length visited by forward >= walked")
print("zero-mask fraction of walked entries", float((m == 0).float().mean()))
print("mean strips evaluated per walked entry", float(pc.float().mean()), "per non-zero entry", float(pc[m != 0].float().mean()))
print("histogram of popcount", [int((pc == k).sum()) for k in range(5)])
