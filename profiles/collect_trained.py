#!/usr/bin/env python
"""Profiles of the trained / random-camera state (bench.py: trained_random_camera), ON THE GPU BOX, into
gpurun_out/prof_<tag>/:

    python profiles/collect_trained.py <tag> [training steps] [timed steps]

  1. rocprofv3 --kernel-trace --stats -- python bench.py --only-trained ...
       -> kernel_stats.csv (whole process), kernel_windows.json / kernel_windows.txt: per-kernel launches and microseconds
          PER ITERATION inside each timed window of the leg.  bench.py launches mark_visible_kernel -- which no iteration
          contains -- at the window boundaries and prints the tags in launch order (`trace_markers`); the trace is cut there.
  2. rocprofv3 --pmc SQ_* (two passes, render_fwd / render_bwd only; markers included)
       -> sq_windows.json: mean wave-instruction counts per launch and window
  3. walk_statistics.json (written by bench.py through E3DGS_TRAINED_TRACE_DIR): list lengths, walked fraction, touched
     fraction, strips per entry, for the untrained and the trained state.

`python profiles/collect_trained.py --install <tag>` (build container) copies the summaries into profiles/<tag>/.
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
MARK = "mark_visible_kernel"
SQ_PASSES = [
    ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_BUSY_CYCLES"],
    ["SQ_INSTS_VALU_TRANS", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_VALU"],
]


def kname(r):
    return r["Kernel_Name"].split("(")[0]


def run(out, name, extra, train, timed):
    d = os.path.join(out, name)
    cmd = ["rocprofv3"] + extra + ["-d", d, "-o", name, "--output-format", "csv", "--", sys.executable,
                                   os.path.join(ROOT, "bench.py"), "--only-trained", "--trained-steps", str(train),
                                   "--trained-timed", str(timed)]
    env = dict(os.environ, TMPDIR="/tmp", E3DGS_TRAINED_TRACE_DIR=out)
    log = os.path.join(out, name + ".log")
    with open(log, "w") as f:
        rc = subprocess.call(cmd, stdout=f, stderr=subprocess.STDOUT, cwd="/tmp", env=env)
    rec = None
    for line in open(log):
        if line.startswith('{"trained_random_camera"'):
            rec = json.loads(line)["trained_random_camera"]
    files = glob.glob(os.path.join(d, "**", "*.csv"), recursive=True)
    if rc != 0 or not files or rec is None:
        print("pass %s failed (rc %d): see %s" % (name, rc, log))
    return files, rec


def cut(rows, markers):
    """rows in launch order -> {tag: rows between `<tag>:timed:begin` and `<tag>:timed:end`}."""
    pos = [i for i, r in enumerate(rows) if kname(r) == MARK]
    if len(pos) != len(markers):
        print("marker count mismatch: %d launches of %s, %d tags" % (len(pos), MARK, len(markers)))
        return {}
    at = dict(zip(markers, pos))
    out = {}
    for m in markers:
        if m.endswith(":timed:begin"):
            tag = m[:-len(":timed:begin")]
            out[tag] = rows[at[m] + 1:at[tag + ":timed:end"]]
    return out


def collect(tag, train, timed):
    out = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    os.makedirs(out, exist_ok=True)
    from bench import source_fingerprint
    summary = {"source_fingerprint": source_fingerprint(),
               "command": "python bench.py --only-trained --trained-steps %d --trained-timed %d" % (train, timed)}
    files, rec = run(out, "trace", ["--kernel-trace", "--stats"], train, timed)
    if rec is not None:
        json.dump(rec, open(os.path.join(out, "bench_record_under_rocprof.json"), "w"), indent=1)
    stats = [f for f in files if f.endswith("kernel_stats.csv")]
    if stats:
        shutil.copy(stats[0], os.path.join(out, "kernel_stats.csv"))
    windows = {}
    for f in files:
        if f.endswith("kernel_trace.csv") and rec is not None:
            rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
            for wtag, wrows in cut(rows, rec["trace_markers"]).items():
                iters = sum(1 for r in wrows if kname(r) == "render_bwd_kernel")
                acc = collections.defaultdict(lambda: [0, 0.0])
                for r in wrows:
                    a = acc[kname(r)]
                    a[0] += 1
                    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                span = (int(wrows[-1]["End_Timestamp"]) - int(wrows[0]["Start_Timestamp"])) / 1e3 if wrows else 0.0
                windows[wtag] = {
                    "iterations": iters, "wall_us_per_iteration": round(span / max(iters, 1), 1),
                    "kernel_us_per_iteration": round(sum(a[1] for a in acc.values()) / max(iters, 1), 1),
                    "kernels": {k: {"launches_per_iteration": round(a[0] / max(iters, 1), 2),
                                    "us_per_iteration": round(a[1] / max(iters, 1), 2), "avg_us": round(a[1] / a[0], 2)}
                                for k, a in sorted(acc.items(), key=lambda kv: -kv[1][1])}}
    summary["windows"] = windows
    json.dump(windows, open(os.path.join(out, "kernel_windows.json"), "w"), indent=1)
    with open(os.path.join(out, "kernel_windows.txt"), "w") as f:
        f.write("rocprofv3 --kernel-trace -- %s\nmicroseconds per ITERATION inside the timed windows of the leg "
                "(cut at the mark_visible_kernel launches)\n\n" % summary["command"])
        tags = list(windows)
        names = []
        for t in tags:
            for k in windows[t]["kernels"]:
                if k not in names:
                    names.append(k)
        f.write("%-52s" % "kernel" + "".join("%32s" % t[:31] for t in tags) + "\n")
        for k in names:
            f.write("%-52s" % k[:51] + "".join("%32.1f" % windows[t]["kernels"].get(k, {}).get("us_per_iteration", 0.0)
                                              for t in tags) + "\n")
        f.write("%-52s" % "sum of kernels" + "".join("%32.1f" % windows[t]["kernel_us_per_iteration"] for t in tags) + "\n")
        f.write("%-52s" % "wall (first start .. last end)" + "".join("%32.1f" % windows[t]["wall_us_per_iteration"] for t in tags) + "\n")
        f.write("%-52s" % "iterations" + "".join("%32d" % windows[t]["iterations"] for t in tags) + "\n")
    # ---- instruction mix of the compositing kernels per window (short windows: counters serialise the kernels)
    sq = collections.defaultdict(lambda: collections.defaultdict(dict))
    for i, ctrs in enumerate(SQ_PASSES):
        files, rec_i = run(out, "sq%d" % i, ["--pmc"] + ctrs + ["--kernel-include-regex",
                                                              "render_(fwd|bwd)_kernel|" + MARK], train, 40)
        if rec_i is None:
            continue
        for f in files:
            if not f.endswith("counter_collection.csv"):
                continue
            rows = list(csv.DictReader(open(f)))
            # one row per (dispatch, counter): rebuild dispatch order
            disp = collections.OrderedDict()
            for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
                disp.setdefault(int(r["Dispatch_Id"]), {"Kernel_Name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
            for wtag, wrows in cut(list(disp.values()), rec_i["trace_markers"]).items():
                for kn in ("render_fwd_kernel", "render_bwd_kernel"):
                    sel = [r for r in wrows if kname(r) == kn]
                    for c in ctrs:
                        vals = [r[c] for r in sel if c in r]
                        if vals:
                            sq[wtag][kn][c] = int(sum(vals) / len(vals))
                    sq[wtag][kn]["launches"] = len(sel)
        for wtag in sq:
            if rec_i.get(wtag):
                sq[wtag]["tile_instances_3views_mean"] = rec_i[wtag].get("tile_instances_3views_mean")
    summary["sq_windows"] = sq
    json.dump(sq, open(os.path.join(out, "sq_windows.json"), "w"), indent=1)
    json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
    print(open(os.path.join(out, "kernel_windows.txt")).read())


def install(tag):
    src, dst = os.path.join(ROOT, "gpurun_out", "prof_" + tag), os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    for f in ("kernel_stats.csv", "kernel_windows.json", "kernel_windows.txt", "sq_windows.json", "walk_statistics.json",
              "bench_record_under_rocprof.json", "summary.json"):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    print("installed", dst)


if __name__ == "__main__":
    if sys.argv[1] == "--install":
        install(sys.argv[2])
    else:
        collect(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1000, int(sys.argv[3]) if len(sys.argv) > 3 else 200)
