import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from event_3dgs_amd import synth, densify
from event_3dgs_amd.cameras import orbit_camera
from event_3dgs_amd.train_step import EventTrainer
dev = torch.device("cuda", 0)
N, W, H = 1_000_000, 1920, 1080
params = synth.make_scene(N, "trained", seed=0, device=dev)
cams = [orbit_camera(0, 64, W, H, device=dev, daz=d) for d in (0.0, 0.005, 0.015)]
bg = torch.zeros(3, device=dev)
gp = dict(params); gp["xyz"] = params["xyz"] + 0.01 * torch.randn(N, 3, generator=torch.Generator().manual_seed(1)).to(dev)
gt = EventTrainer(gp, dev)
gts = [(torch.round(gt.render_raw(c, bg)["color"].clamp(0, 1) * 255) / 255).contiguous() for c in cams]
del gt
tr = EventTrainer(params, dev, track_densification_stats=True)
stats = densify.DensifyStats(tr.N, dev)
for rep in range(3):
    for _ in range(5):
        tr.step(*cams, *gts, bg)
        stats.update(tr.viewspace_grad, tr.last_radii)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = tr.densify_and_prune(stats, 0.0002, 0.005, 4.4, 20, 0.01)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if stats.denom.shape[0] != tr.N:
        stats = densify.DensifyStats(tr.N, dev)
    print("densify_and_prune: %.1f ms, N = %d" % ((t1 - t0) * 1e3, n))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step(*cams, *gts, bg); torch.cuda.synchronize()
    print("  first step after: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
