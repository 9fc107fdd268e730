"""bench.py's launch contract: `python bench.py --gpus N` must not exit at argument parsing (it re-launches itself
under torch.distributed.run, one rank per GPU) and rank 0 prints exactly one JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_builds_the_torchrun_command(monkeypatch):
    """CPU: the launcher re-runs the same arguments under torch.distributed.run on 127.0.0.1."""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd: seen.setdefault("cmd", cmd) and 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    assert bench.self_launch(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and cmd[-5].endswith("bench.py")


def test_mismatched_world_size_is_an_error():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("force_fallback,schedule", [(False, "allreduce"), (False, "rs_ag"), (True, "allreduce")])
def test_bench_gpus2_self_launches_and_prints_one_json_line(force_fallback, schedule):
    """The driver's command form on a one-GPU box: both ranks share cuda:0 and talk over gloo (test hooks of bench.py)."""
    env = dict(os.environ, E3DGS_BENCH_BACKEND="gloo", E3DGS_BENCH_DEVICE="0", E3DGS_DP_SCHEDULE=schedule)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    if force_fallback:
        env["E3DGS_BENCH_FORCE_FALLBACK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "tiny", "--steps", "3",
                        "--warmup", "1"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size"] == 2 and out["comm_backend"] == "gloo"
    # (the census over the backend's own all-gather: two ranks, ONE device, and no RCCL rank at all under gloo)
    assert out["rccl_ranks"] == 0 and out["distinct_devices_in_the_communicator"] == 1
    assert out["scaling"] == "weak" and out["value"] > 0
    assert out["dp_fallback"] is force_fallback
    assert out["config"]["workload"] == "tiny"
    # the exchange is diagnosable from the line alone
    comm = out["comm"]
    for key in ("exposed_comm_ms", "allreduce_ms", "allgather_ms", "reduce_scatter_plus_allgather_ms", "local_step_ms"):
        assert isinstance(comm[key], float), key
    assert comm["nonsh_schedule"] == schedule and "NCCL_ALGO" in comm["env"] and "NCCL_PROTO" in comm["env"]
    assert comm["sh_exchange"] == ("allreduce" if force_fallback else "factorised")


@pytest.mark.gpu
def test_bench_gpus8_dry_run_on_one_gpu():
    """`bench.py --gpus 8` end to end before an 8-GPU node ever sees it: eight ranks share cuda:0 over gloo (the test hooks
    above), tiny workload -- the launch, the world-8 exchange (factorised SH all-gather of 8 blocks, all-reduce of the
    non-SH groups), the max-over-ranks timing and the one JSON line of rank 0."""
    env = dict(os.environ, E3DGS_BENCH_BACKEND="gloo", E3DGS_BENCH_DEVICE="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "tiny", "--steps", "3",
                        "--warmup", "1"], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["world_size"] == 8 and out["comm_backend"] == "gloo"
    assert out["rccl_ranks"] == 0 and out["distinct_devices_in_the_communicator"] == 1
    assert out["scaling"] == "weak" and out["value"] > 0 and out["dp_fallback"] is False
    assert out["comm"]["sh_exchange"] == "factorised"


@pytest.mark.gpu
def test_bench_strict_mode_refuses_anything_but_one_rccl_rank_per_gpu():
    """E3DGS_BENCH_STRICT=1 (what a real multi-GPU run sets): ranks that share a device or talk over another backend, and a
    failing first exchange, end the run with a message instead of a quieter, slower line."""
    env = dict(os.environ, E3DGS_BENCH_BACKEND="gloo", E3DGS_BENCH_DEVICE="0", E3DGS_BENCH_STRICT="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "tiny", "--steps", "2",
                        "--warmup", "1"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode != 0
    assert "E3DGS_BENCH_STRICT=1" in (r.stderr + r.stdout) and "one RCCL rank per GPU" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 4, 8])
def test_bench_real_rccl_ranks_one_per_gpu_strict(n):
    """The driver's multi-GPU command form with E3DGS_BENCH_STRICT=1 on a box that HAS n GPUs (skipped elsewhere): one RCCL
    rank per GPU, no fallback, `rccl_ranks` == n from the communicator's own all-gather."""
    import torch
    if torch.cuda.device_count() < n:
        pytest.skip(f"needs {n} GPUs (this box has {torch.cuda.device_count()})")
    env = dict(os.environ, E3DGS_BENCH_STRICT="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "E3DGS_BENCH_BACKEND", "E3DGS_BENCH_DEVICE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--config", "tiny", "--steps", "3",
                        "--warmup", "1"], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == n and out["rccl_ranks"] == n and out["comm_backend"] == "nccl" and out["dp_fallback"] is False
    assert out["distinct_devices_in_the_communicator"] == n and out["value"] > 0


@pytest.mark.gpu
def test_bench_single_gpu_line_carries_the_contract():
    """`python bench.py` (N = 1; the tiny workload to keep the test short): one JSON line with the contract's keys, the
    `roofline` object of the dominant kernel measured with HIP events in the timed region, the `cpu_baseline` of the
    oracle, and the extra measurements that never enter `value`."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "tiny", "--steps", "6", "--warmup", "2"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert out["metric"] == base["metric"] and out["unit"] == "iters/s"
    assert out["n_gpus"] == 1 and out["steps"] == 6 and out["warmup"] == 2 and out["higher_is_better"] is True
    assert out["scaling"] == "weak" and out["vs_baseline"] is None and out["dtype"] == "f32" and out["data"] == "synthetic"
    assert abs(out["value"] * out["ms_per_step"] / 1e3 - 1.0) < 1e-2          # value = iterations / s of the timed region
    assert out["config"]["workload"] == "tiny" and "model" not in out["config"]
    rf = out["roofline"]
    assert rf["bound"] in ("hbm", "mfma", "valu") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert rf["achieved"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and "traffic" in rf
    assert rf["kernel"] == "render_bwd_kernel" and rf["avg_launch_ms"] > 0
    cb = out["cpu_baseline"]
    assert cb["value"] > 0 and cb["unit"] == "iters/s" and cb["cores"] >= 1 and cb["kind"] in ("port", "reference")
    assert isinstance(cb["sample"], str) and cb["sample"]
    assert out["device_allocs_in_timed_region"] == 0
    for key in ("contrast_only_substep", "shared_pose_iteration", "dropin_autograd_step", "trained_random_camera"):
        assert key in out
    trc = out["trained_random_camera"]
    assert trc["trained_random_camera"]["ms_per_step"] > 0 and trc["trained_random_camera"]["count_retries"] >= 0
    assert trc["walk_statistics"]["trained"]["intensity"]["walked_fraction"] > 0
    assert len(trc["trace_markers"]) == 10
    assert out["shared_pose_iteration"]["taken"] is True and out["shared_pose_iteration"]["renders"] == 2
    # how to read the headline: a sustained window, a frozen workload, the literal reference binning
    assert out["sustained"]["steps"] >= 500 and out["sustained"]["ms_per_step"] > 0
    assert out["static_workload"]["ms_per_step"] > 0 and out["static_workload"]["parameters_moved_by"] == 0.0
    rb = out["reference_binning_iteration"]
    assert rb["ms_per_step"] > 0 and rb["tile_instances_reference_binning_3views"] >= rb["tile_instances_exact_culling_3views"]
    assert out["config"]["tile_instances_reference_binning"] == rb["tile_instances_reference_binning"]
    assert out["config"]["tile_instances_reference_binning"] >= out["config"]["tile_instances"]
    assert rf["frac_on_reference_instances"] >= rf["frac"]
    # the adoption ladder: every rung measured, the last one on torch.optim's interface with this repo's pieces
    lad = out["dropin_autograd_step"]["tiny"]["ladder"]
    for key in ("rung0_unmodified", "rung1_render", "rung2_render_loss", "rung3_render_loss_optimizer",
                "rung4_one_call_for_the_three_renders"):
        assert lad[key]["ms"] > 0 and lad[key]["rasteriser_kernels_ms"] > 0, key
        assert lad[key]["render_bwd_ms"] > 0 and lad[key]["geom_bwd_ms"] > 0, key      # backward kernels timed (autograd thread)
    assert out["dropin_autograd_step"]["tiny"]["cpp_autograd_node"] is True
