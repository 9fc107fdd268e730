"""The distributed schedules of EventTrainer over the REAL RCCL backend, on the one GPU a development box has:
`dist.init_process_group("nccl", world_size=1, device_id=...)` drives the true RCCL entry points (ncclAvg all-reduce,
all_gather_into_tensor, reduce_scatter_tensor) with a single rank, and EventTrainer(force_distributed=True) takes every
branch a multi-rank run takes (`dist_on`): both non-SH schedules, the factorised SH all-gather and the side-stream overlap.
With one rank the mean over the ranks is the identity, so each schedule must leave the parameters and the Adam moments
BIT-IDENTICAL to the local step.  (Two-rank semantics are covered over gloo: tests/test_hip_multirank.py.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SCRIPT = r'''
import json, os, sys
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29631")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from event_3dgs_amd import synth, parallel
from event_3dgs_amd.cameras import orbit_camera
from event_3dgs_amd.train_step import EventTrainer
N, W, H = 20000, 320, 240
params = synth.make_scene(N, "trained", seed=0, device=dev)
cams = [orbit_camera(0, 16, W, H, device=dev, daz=d) for d in (0.0, 0.004, 0.012)]
bg = torch.zeros(3, device=dev)
gp = dict(params); gp["xyz"] = params["xyz"] + 0.01 * torch.randn(N, 3, generator=torch.Generator().manual_seed(1)).to(dev)
gt = EventTrainer(gp, dev)
gts = [gt.render_raw(c, bg)["color"].clone() for c in cams]
def run(**kw):
    tr = EventTrainer(params, dev, **kw)
    losses = [float(tr.step(*cams, *gts, bg)[0]) for _ in range(3)]
    tr.sync_features(); tr.sync_optimizer_state(); torch.cuda.synchronize()
    return tr, losses
ref, l_ref = run()                                        # the local step (no collective)
assert not ref.multi
out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "cases": {}}
cases = {
    "allreduce+factorised+overlap": dict(force_distributed=True, dp_schedule="allreduce", factorize_sh=True, overlap_features=True),
    "allreduce+factorised": dict(force_distributed=True, dp_schedule="allreduce", factorize_sh=True, overlap_features=False),
    "allreduce+sh_allreduce": dict(force_distributed=True, dp_schedule="allreduce", factorize_sh=False, overlap_features=False),
    "rs_ag+factorised+overlap": dict(force_distributed=True, dp_schedule="rs_ag", factorize_sh=True, overlap_features=True),
    "rs_ag+sh_allreduce": dict(force_distributed=True, dp_schedule="rs_ag", factorize_sh=False, overlap_features=False),
}
for name, kw in cases.items():
    tr, l = run(**kw)
    assert tr.multi and parallel.FORCE_SINGLE_RANK_COLLECTIVES and tr.dp_schedule == kw["dp_schedule"]
    out["cases"][name] = {
        "params_equal": bool(torch.equal(tr.flat, ref.flat)), "exp_avg_equal": bool(torch.equal(tr.exp_avg, ref.exp_avg)),
        "exp_avg_sq_equal": bool(torch.equal(tr.exp_avg_sq, ref.exp_avg_sq)), "loss_equal": l == l_ref,
        "max_param_diff": float((tr.flat - ref.flat).abs().max()), "finite": bool(torch.isfinite(tr.flat).all())}
# the collectives alone, on their real buffer shapes: ncclAvg all-reduce, all-gather, reduce-scatter + all-gather
x = torch.randn(11 * N + 1, device=dev); y = x.clone()
parallel.allreduce_mean_(x); out["allreduce_avg_identity"] = bool(torch.equal(x, y))
g = torch.zeros(1, 9 * N + 9, device=dev); c = torch.randn(9 * N + 9, device=dev)
parallel.allgather_async_(g, c).wait(); out["allgather_identity"] = bool(torch.equal(g[0], c))
sh = torch.zeros(x.numel(), device=dev)
parallel.reduce_scatter_mean_async_(sh, y).wait(); out["reduce_scatter_identity"] = bool(torch.equal(sh, y))
torch.cuda.synchronize()
print("RESULT " + json.dumps(out))
dist.destroy_process_group()
'''


def test_distributed_schedules_over_rccl_with_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + SCRIPT], env=env, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, r.stdout[-2000:]
    out = json.loads(line[0][7:])
    print(out)
    assert out["backend"] == "nccl" and out["world"] == 1            # "nccl" IS RCCL on ROCm
    assert out["allreduce_avg_identity"] and out["allgather_identity"] and out["reduce_scatter_identity"]
    for name, c in out["cases"].items():
        assert c["finite"], name
        assert c["loss_equal"], (name, c)
        assert c["params_equal"] and c["exp_avg_equal"] and c["exp_avg_sq_equal"], (name, c)
