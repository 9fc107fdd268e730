"""The committed profile summary bench.py decorates its line with (profiles/traffic.json = profiles/r06_final/summary.json,
written by profiles/collect.py on the GPU box) is complete: the per-iteration attribution found the iteration's kernels
and the dominant kernel's HBM traffic is there under the name bench.py looks up.  (Whether it still describes the
current sources is reported by bench.py itself: `traffic_stale`.)"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_profile_summary_is_complete():
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert len(t["source_fingerprint"]) == 16
    per_iter = t["kernels_per_iteration"]
    assert len(per_iter) >= 20, sorted(per_iter)                       # one event iteration launches ~23 distinct kernels
    for name in ("render_bwd_kernel", "render_fwd_kernel", "run_reduce_kernel", "sh_adam_views_kernel"):
        hits = [k for k in per_iter if name in k]          # (template instances: "void sh_adam_views_kernel<true, 3>")
        assert len(hits) == 1 and per_iter[hits[0]]["launches_per_iteration"] == 1, (name, hits)
    tr = t["traffic_per_kernel"]["render_bwd_kernel"]
    assert tr["hbm_bytes_per_launch"] == tr["fetch_bytes"] + tr["write_bytes"] > 1e9
    assert set(t["stages"]) >= {"preprocess", "sort_depth", "scan_emit", "sort_tile", "render_fwd", "render_bwd", "geom_bwd",
                                "optimizer", "event_loss"}
    assert 1500.0 < t["iteration_us_sum_of_kernels"] < 5000.0
