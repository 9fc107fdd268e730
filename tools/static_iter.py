#!/usr/bin/env python
"""The benchmark iteration WITHOUT its optimizer step, repeated on fixed parameters (forward, event loss, backward of the
three views): a static workload for same-box comparisons of kernel variants whose results differ (timing experiments that
would otherwise let the scene drift apart).  Usage: python tools/static_iter.py [iterations] [N] [W] [H]
Run under `rocprofv3 --kernel-trace --stats` (tools/static_kcmp.sh) or read the wall time it prints."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_3dgs_amd import synth
from event_3dgs_amd.cameras import orbit_camera
from event_3dgs_amd.train_step import EventTrainer
dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
H = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
params = synth.make_scene(N, "trained", seed=0, device=dev)
cams = [orbit_camera(0, 64, W, H, device=dev, daz=d) for d in (0.0, 0.005, 0.015)]
bg = torch.zeros(3, device=dev)
gt = EventTrainer(params, dev)
gts = [(torch.round(gt.render_raw(c, bg)["color"].clamp(0, 1) * 255.0) / 255.0).contiguous() for c in cams]
del gt
tr = EventTrainer(params, dev)
step = lambda: tr.compute_gradients(*cams, *gts, bg, sh_via_colour=True)
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    step()
torch.cuda.synchronize()
print("STATIC_ITER %d x (forward + loss + backward, no optimizer) N=%d %dx%d: %.4f ms per iteration"
      % (K, N, W, H, (time.perf_counter() - t0) / K * 1e3))
