#!/usr/bin/env python
"""Collects the profiles the bench line is read against, ON THE GPU BOX, into gpurun_out/prof_<tag>/:

    python profiles/collect.py <tag> [git head]

  1. rocprofv3 --kernel-trace --stats          -> kernel_stats.csv, kernel_stats_top.txt   (per-kernel durations)
  2. rocprofv3 --pmc FETCH_SIZE                 \\  separate passes, every kernel of the iteration
  3. rocprofv3 --pmc WRITE_SIZE                 /  -> pmc_traffic_all_kernels.txt
  4. rocprofv3 --pmc SQ_* (three passes)        -> sq_instruction_mix.txt               (the two compositing kernels)
  -> summary.json: HBM-side bytes per launch of every kernel and per bench.py stage, the instruction mix of the compositing
     kernels, and the FINGERPRINT of the sources the profiled library was built from (bench.py compares it with the
     sources it runs on and marks the numbers stale when they differ).

The duration pass profiles the driver's window, `python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-substep`; the
counter passes `--steps 4 --warmup 2` (the 3-view launches of the benchmark iteration are the largest launches of each
kernel: max over launches).  PMC passes carry no trace domain
besides kernel dispatch.  gfx950 corrections (MI355X_MICROARCH.md, re-calibrated with tools/ubench/pmc_calib.hip):
FETCH_SIZE counts 128-byte line requests at 64 bytes -> x2; WRITE_SIZE x1.  Afterwards, in the build container:
`python profiles/collect.py --install <tag>` copies the summaries into profiles/<tag>/ and profiles/traffic.json.
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STAGES = {          # bench.py stage -> kernel-name prefixes (rocprof names; template arguments included)
    "preprocess": ["void preprocess_kernel"],
    "sort_depth": ["void radix_hist_kernel<unsigned int", "void radix_scatter_kernel<unsigned int"],
    "scan_emit": ["void bin_kernel<false>", "void bin_kernel<true>", "void scan_chained_kernel<true>", "bin_fused_kernel"],
    "sort_tile": ["void radix_hist_kernel<unsigned short", "void radix_hist_seg_kernel<unsigned short",
                  "void radix_scatter_kernel<unsigned short"],
    "tile_ranges": ["tile_order_reg_kernel", "void tile_ranges_kernel"],
    "render_fwd": ["render_fwd_kernel"],
    "render_bwd": ["render_bwd_kernel"],
    "geom_bwd": ["run_reduce_kernel", "geom_bwd_multi_kernel", "void geom_bwd_multi_kernel"],
    "optimizer": ["sh_adam_views_kernel", "void sh_adam_views_kernel", "adam_segments_kernel"],
    "event_loss": ["event_reduce_kernel", "event_finalize_kernel", "event_grad_kernel", "event_fused_kernel"],
}
SQ_PASSES = [
    ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU",
     "SQ_INSTS_SALU", "SQ_INSTS_LDS"],
    ["SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT",
     "SQ_WAIT_INST_LDS", "SQ_INSTS_BRANCH"],
    ["SQ_INSTS_VALU_TRANS", "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_LDS_IDX_ACTIVE"],
]


def kname(r):
    return r["Kernel_Name"].split("(")[0]


def rocprof(out, name, extra, steps=4, warmup=2):
    d = os.path.join(out, name)
    cmd = ["rocprofv3"] + extra + ["-d", d, "-o", name, "--output-format", "csv", "--", sys.executable,
                                   os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", str(warmup),
                                   "--no-cpu-baseline", "--no-substep"]
    with open(os.path.join(out, name + ".log"), "w") as log:
        rc = subprocess.call(cmd, stdout=log, stderr=subprocess.STDOUT, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
    files = glob.glob(os.path.join(d, "**", "*.csv"), recursive=True)
    if rc != 0 or not files:
        print("rocprofv3 pass %s failed (rc %d): see %s.log" % (name, rc, name))
    return files


def max_per_kernel(files, suffix, counter=None):
    """Per kernel: the counter of its TYPICAL largest launch -- the median over the launches with the kernel's largest grid.
    The largest grid selects the 3-view launches of the training iterations (the ground-truth renders in front of them are
    single-view launches of the same kernels); the median among those selects a training iteration (9 of the 13 launches of
    `bench.py --steps 4 --warmup 2`) rather than the general-form backward of bench.py's in-kernel clock measurement, which
    shares the grid but not the rank-1 body the timed iterations run."""
    per = collections.defaultdict(list)
    for f in files:
        if not f.endswith(suffix):
            continue
        for r in csv.DictReader(open(f)):
            if counter and r["Counter_Name"] != counter:
                continue
            per[kname(r)].append((int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"])))
    agg = {}
    for k, v in per.items():
        g = max(x[0] for x in v)
        vals = sorted(x[1] for x in v if x[0] == g)
        agg[k] = vals[len(vals) // 2]
    return agg


def collect(tag, head):
    from bench import source_fingerprint
    out = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    os.makedirs(out, exist_ok=True)
    summary = {"source_fingerprint": source_fingerprint(), "git_head": head,
               "command": "python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-substep (kernel trace); --steps 4 "
                          "--warmup 2 (one rocprofv3 counter pass each)"}
    # ---- 1. kernel durations
    # (the duration pass runs the DRIVER's window -- 5 warm-up + 20 timed iterations -- so that the kernel's average is
    # taken over the launches the bench line's HIP events bracket, not over a cold 4-step run)
    files = rocprof(out, "trace", ["--kernel-trace", "--stats"], steps=20, warmup=5)
    stats = [f for f in files if f.endswith("kernel_stats.csv")]
    if stats:
        shutil.copy(stats[0], os.path.join(out, "kernel_stats.csv"))
        rows = list(csv.DictReader(open(stats[0])))
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        with open(os.path.join(out, "kernel_stats_top.txt"), "w") as f:
            f.write("rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-substep\n(one launch of every rasteriser kernel covers the 3 "
                    "views of an event iteration; the single-view launches in the min column come from the ground-truth "
                    "renders bench.py makes before the timed region)\n\n")
            for r in rows[:32]:
                f.write(f"{r['Name'].split('(')[0][:64]:64s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} "
                        f"min_us={float(r['MinNs'])/1e3:9.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f} "
                        f"{100*float(r['TotalDurationNs'])/tot:5.1f}%\n")
    # per-dispatch durations of ONE training iteration: the dispatches between two consecutive render_bwd_kernel launches
    # (an interior window of the timed steps; the ground-truth renders in front and the statistics renders behind the
    # training loop launch the per-Gaussian kernels with the same grids and must not be counted)
    durations = {}
    for f in files:
        if f.endswith("kernel_trace.csv"):
            rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
            marks = [i for i, r in enumerate(rows) if kname(r) == "render_bwd_kernel"]
            windows = [rows[marks[j] + 1:marks[j + 1] + 1] for j in range(1, len(marks) - 1)]       # interior iterations
            if not windows:
                continue
            per = collections.defaultdict(list)             # kernel -> [(launches, total us)] per window
            for w in windows:
                acc = collections.defaultdict(lambda: [0, 0.0])
                for r in w:
                    a = acc[kname(r)]
                    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                for k, a in acc.items():
                    per[k].append(tuple(a))
            for k, v in per.items():
                if len(v) * 2 < len(windows):
                    continue                                 # not part of every iteration
                n_l = sorted(x[0] for x in v)[len(v) // 2]
                us = [x[1] for x in v if x[0] == n_l]
                durations[k] = {"launches_per_iteration": n_l, "us_per_iteration": round(sum(us) / len(us), 2)}
            summary["iterations_in_window_stats"] = len(windows)
            summary["iteration_us_sum_of_kernels"] = round(sum(d["us_per_iteration"] for d in durations.values()), 1)
    summary["kernels_per_iteration"] = durations
    # ---- 2./3. HBM-side traffic of every kernel
    fetch = max_per_kernel(rocprof(out, "fetch", ["--pmc", "FETCH_SIZE"]), "counter_collection.csv", "FETCH_SIZE")
    write = max_per_kernel(rocprof(out, "write", ["--pmc", "WRITE_SIZE"]), "counter_collection.csv", "WRITE_SIZE")
    traffic = {k: {"fetch_bytes": int(2 * fetch.get(k, 0.0) * 1024), "write_bytes": int(write.get(k, 0.0) * 1024)}
               for k in sorted(set(fetch) | set(write))}
    for v in traffic.values():
        v["hbm_bytes_per_launch"] = v["fetch_bytes"] + v["write_bytes"]
    summary["traffic_per_kernel"] = traffic
    with open(os.path.join(out, "pmc_traffic_all_kernels.txt"), "w") as f:
        f.write("%-56s %12s %12s   (MB per launch: median over the launches with the kernel's largest grid; FETCH_SIZE x2 = gfx950 correction)\n" % ("kernel", "fetch_MB", "write_MB"))
        for k, v in sorted(traffic.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
            f.write("%-56s %12.1f %12.1f\n" % (k[:56], v["fetch_bytes"] / 1e6, v["write_bytes"] / 1e6))
    # bytes per ITERATION and bench.py stage: per kernel, bytes of its largest launch x its launches per iteration (the
    # exclusive chained scans serve both sorts -- four launches per iteration belong to the depth sort, two to the tile
    # sort; their few MB are booked at the size of the largest one)
    stage = {}
    for name, prefixes in STAGES.items():
        ks = [k for k in traffic if any(k.startswith(p) for p in prefixes) and k in durations]
        b = sum(traffic[k]["hbm_bytes_per_launch"] * durations[k]["launches_per_iteration"] for k in ks)
        us = sum(durations[k]["us_per_iteration"] for k in ks)
        scan = "void scan_chained_kernel<false>"
        if scan in durations and name in ("sort_depth", "sort_tile"):
            share = {"sort_depth": 4, "sort_tile": 2}[name] / max(durations[scan]["launches_per_iteration"], 1)
            b += traffic.get(scan, {}).get("hbm_bytes_per_launch", 0) * durations[scan]["launches_per_iteration"] * share
            us += durations[scan]["us_per_iteration"] * share
        stage[name] = {"bytes_per_iteration": int(b), "kernel_us_per_iteration": round(us, 1), "kernels": ks}
    summary["stages"] = stage
    # ---- 4. instruction mix of the compositing kernels
    mix = collections.defaultdict(dict)
    for i, ctrs in enumerate(SQ_PASSES):
        files = rocprof(out, "sq%d" % i, ["--pmc"] + ctrs + ["--kernel-include-regex", "render_(fwd|bwd)_kernel"])
        for c in ctrs:
            for k, v in max_per_kernel(files, "counter_collection.csv", c).items():
                mix[k][c] = int(v)
    summary["sq"] = mix
    with open(os.path.join(out, "sq_instruction_mix.txt"), "w") as f:
        for k in sorted(mix):
            for c in sorted(mix[k]):
                f.write("%-24s %-24s %d\n" % (k[:24], c, mix[k][c]))
    json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
    print(open(os.path.join(out, "kernel_stats_top.txt")).read() if stats else "no kernel stats")
    print(json.dumps({k: v for k, v in summary.items() if k in ("source_fingerprint", "git_head")}))


def install(tag):
    src, dst = os.path.join(ROOT, "gpurun_out", "prof_" + tag), os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    for f in ("kernel_stats.csv", "kernel_stats_top.txt", "pmc_traffic_all_kernels.txt", "sq_instruction_mix.txt", "summary.json",
              "ubench.txt", "ubench_valu_rate.txt", "ubench_mixed_issue.txt"):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    shutil.copy(os.path.join(src, "summary.json"), os.path.join(ROOT, "profiles", "traffic.json"))
    print("installed", dst, "and profiles/traffic.json")


if __name__ == "__main__":
    if sys.argv[1] == "--install":
        install(sys.argv[2])
    else:
        collect(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "unknown")
