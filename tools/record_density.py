#!/usr/bin/env python
"""How the non-zero gradient records are distributed over the record array (slot order): fraction of records that the
compositing backward actually touched, and fraction of 128-byte lines / 64-byte sectors that hold at least one."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_3dgs_amd import _lib, rasterizer, synth
from event_3dgs_amd.cameras import orbit_camera
from event_3dgs_amd.train_step import EventTrainer
dev = torch.device("cuda", 0)
N, W, H = 1_000_000, 1920, 1080
params = synth.make_scene(N, "trained", seed=0, device=dev)
cams = [orbit_camera(0, 64, W, H, device=dev, daz=d) for d in (0.0, 0.005, 0.015)]
bg = torch.zeros(3, device=dev)
gp = dict(params); gp["xyz"] = params["xyz"] + 0.01 * torch.randn(N, 3, generator=torch.Generator().manual_seed(1)).to(dev)
gt = EventTrainer(gp, dev)
gts = [gt.render_raw(c, bg)["color"].clone() for c in cams]
tr = EventTrainer(params, dev)
cap = {}
orig = rasterizer.backward_multi
def spy(raw, dpix, out, flags=None, grad_acc=None):
    cap["raw"] = raw
    return orig(raw, dpix, out, flags, grad_acc)
rasterizer.backward_multi = spy
tr.compute_gradients(*cams, *gts, bg, sh_via_colour=True)
torch.cuda.synchronize()
raw = cap["raw"]
I = int(raw["num_rendered"])
acc = raw["pool"].typed("grad_acc", (I + 3 * N, _lib.ACC_STRIDE)).reshape(-1)[:9 * I].reshape(I, 9)
touched = (acc != 0).any(1)
print("instances %d, touched %.1f %%" % (I, 100 * touched.float().mean().item()))
first = torch.arange(I, device=dev) * 36          # byte offset of every record
for gran in (128, 64):
    lo, hi = first // gran, (first + 35) // gran
    nl = int(hi.max().item()) + 1
    occ = torch.zeros(nl, dtype=torch.bool, device=dev)
    occ[lo[touched]] = True; occ[hi[touched]] = True
    print("%d-byte granules holding a touched record: %.1f %%" % (gran, 100 * occ.float().mean().item()))
