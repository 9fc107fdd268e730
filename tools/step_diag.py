import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from event_3dgs_amd import synth
from event_3dgs_amd.cameras import orbit_camera
from event_3dgs_amd.train_step import EventTrainer
dev = torch.device("cuda", 0)
N, W, H = 1_000_000, 1920, 1080
params = synth.make_scene(N, "trained", seed=0, device=dev)
cams = [orbit_camera(0, 64, W, H, device=dev, daz=d) for d in (0.0, 0.005, 0.015)]
bg = torch.zeros(3, device=dev)
gt = EventTrainer(params, dev)
gts = [gt.render_raw(c, bg)["color"].clone() for c in cams]
tr = EventTrainer(params, dev)
ts = []
for i in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step(*cams, *gts, bg)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((round((t1 - t0) * 1e3, 2), round((t2 - t0) * 1e3, 2)))
print("host_ms/total_ms per step:", ts)
print(torch.cuda.memory_stats()["num_alloc_retries"], torch.cuda.memory_stats()["num_device_alloc"], torch.cuda.memory_stats()["num_device_free"])
