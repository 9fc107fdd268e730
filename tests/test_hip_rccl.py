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


# ---------------------------------------------------------------------------------------------------------------
# Real RCCL ranks, one per GPU: collected everywhere, skipped where the box has fewer GPUs than ranks -- the first box with
# two (four, eight) GPUs runs the schedules below without anybody having to remember to.  Same workload, same emulation and
# the same tolerances as tests/test_hip_multirank.py::test_world_4_and_8_... (gloo, all ranks on one GPU), so "equal to the
# emulation" here and there means RCCL and gloo agree with each other to the emulation's tolerance.
N_RCCL, W_RCCL, H_RCCL = 20003, 208, 144


def _rccl_inputs(rank, dev):
    import torch
    from event_3dgs_amd import synth
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer
    params = synth.make_scene(N_RCCL, "trained", seed=0, device=dev)
    cams = [orbit_camera(3 * rank, 32 if rank >= 5 else 16, W_RCCL, H_RCCL, device=dev, daz=d) for d in (0.0, 0.004, 0.012)]
    bg = torch.zeros(3, device=dev)
    gp = dict(params)
    gp["xyz"] = params["xyz"] + 0.01 * torch.randn(N_RCCL, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    gt = EventTrainer(gp, dev)
    gts = [(torch.round(gt.render_raw(c, bg)["color"].clamp(0, 1) * 255.0) / 255.0).contiguous() for c in cams]
    return params, cams, gts, bg


def _rccl_worker(rank, world, port, out, schedule, overlap, factorize, steps):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["E3DGS_FACTORIZE_SH"] = "1" if factorize else "0"
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from event_3dgs_amd.train_step import EventTrainer
    params, cams, gts, bg = _rccl_inputs(rank, dev)
    tr = EventTrainer(params, dev, overlap_features=overlap, dp_schedule=schedule)
    assert tr.world == world and tr.rank == rank and tr.multi and tr.dp_schedule == schedule
    assert tr.factorize_sh == factorize and tr.overlap_features == overlap and dist.get_backend() == "nccl"
    for _ in range(steps):
        tr.step(*cams, *gts, bg)
    tr.sync_features(); tr.sync_optimizer_state()
    torch.cuda.synchronize()
    torch.save({"flat": tr.flat.cpu(), "m": tr.exp_avg.cpu(), "v": tr.exp_avg_sq.cpu()}, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:          # noqa: BLE001
        return 0


@pytest.mark.parametrize("schedule,overlap,factorize", [("allreduce", True, True), ("rs_ag", True, True),
                                                         ("allreduce", False, False), ("rs_ag", False, False)])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_ranks_one_per_gpu_replicas_identical_and_equal_to_the_mean_of_the_ranks(tmp_path, world, schedule, overlap,
                                                                                      factorize):
    if _gpu_count() < world:
        pytest.skip(f"needs {world} GPUs (this box has {_gpu_count()}); world 1 over RCCL and world 2 / 4 / 8 over gloo on one "
                    "GPU run in test_distributed_schedules_over_rccl_with_one_rank / tests/test_hip_multirank.py")
    import socket
    import torch
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    steps = 2
    out = str(tmp_path / "rank")
    mp.spawn(_rccl_worker, args=(world, port, out, schedule, overlap, factorize, steps), nprocs=world, join=True)
    rs = [torch.load(f"{out}.{r}") for r in range(world)]
    for r in range(1, world):
        for k in ("flat", "m", "v"):
            assert torch.equal(rs[0][k], rs[r][k]), (r, k)                 # replicas bit-identical
    # single-process emulation on GPU 0: the mean of the ranks' gradient buffers, one Adam
    from event_3dgs_amd.train_step import EventTrainer
    dev = torch.device("cuda", 0)
    ins = [_rccl_inputs(r, dev) for r in range(world)]
    bg = ins[0][3]
    ts = [EventTrainer(i[0], dev, overlap_features=False) for i in ins]
    for _ in range(steps):
        for t, (_, cams, gts, _) in zip(ts, ins):
            t.compute_gradients(*cams, *gts, bg)
        mean = torch.stack([t.flat_grad for t in ts]).sum(0).div_(world)
        for t in ts:
            t.flat_grad.copy_(mean)
            t.apply_update()
    torch.cuda.synchronize()
    m_ref, m_got = ts[0].exp_avg.cpu(), rs[0]["m"]
    assert float((m_got - m_ref).norm() / m_ref.norm()) < 1e-4
    v_ref, v_got = ts[0].exp_avg_sq.cpu(), rs[0]["v"]
    assert float((v_got - v_ref).norm() / v_ref.norm()) < 1e-4
    assert float((rs[0]["flat"] - ts[0].flat.cpu()).abs().max()) <= 0.05
