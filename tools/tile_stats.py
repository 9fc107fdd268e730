import os, sys, torch
sys.path.insert(0, os.getcwd())
from event_3dgs_amd import synth, rasterizer
from event_3dgs_amd.cameras import orbit_camera
from event_3dgs_amd.train_step import EventTrainer
dev = torch.device("cuda", 0)
N, W, H = 1_000_000, 1920, 1080
params = synth.make_scene(N, "trained", seed=0, device=dev)
cam = orbit_camera(0, 64, W, H, device=dev)
bg = torch.zeros(3, device=dev)
tr = EventTrainer(params, dev)
raw = tr.render_raw(cam, bg)
st = rasterizer.state_views(raw, N, W, H)
rg = st["ranges"].long()
ln = (rg[:, 1] - rg[:, 0]).float()
nc = st["n_contrib"].float()
print("I", raw["num_rendered"], "tiles", ln.numel())
for q in (0.5, 0.9, 0.99, 0.999, 1.0):
    print("list len q%.3f = %d" % (q, int(torch.quantile(ln, q))))
print("mean len", float(ln.mean()))
# per tile max n_contrib
gx, gy = (W + 15) // 16, (H + 15) // 16
pad = torch.zeros(gy * 16, gx * 16, device=dev); pad[:H, :W] = nc
tmax = pad.view(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(gy * gx, 256).max(dim=1).values
for q in (0.5, 0.9, 0.99, 1.0):
    print("tile max n_contrib q%.2f = %d" % (q, int(torch.quantile(tmax, q))))
print("sum tile max n_contrib", float(tmax.sum()), "sum len", float(ln.sum()))
print("mean n_contrib per pixel", float(nc.mean()))
