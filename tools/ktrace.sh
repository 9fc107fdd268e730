#!/bin/bash
# Per-kernel durations of the benchmark iteration, split by launch size (the depth sort and the tile sort share kernels):
#   bash tools/ktrace.sh <tag> [bench args]   -> gpurun_out/ktrace_<tag>/summary.txt
TAG=${1:-x}; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/ktrace_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace -o trace --output-format csv -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-substep "$@" > $OUT/trace.log 2>&1
python - <<PY > $OUT/summary.txt
import csv, collections, glob
rows = []
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0][:48]
    wg = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1) if "Grid_Size_X" in r else int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
    agg[(name, wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = collections.defaultdict(float)
for (name, wg), v in agg.items():
    tot[name] += sum(v)
print("%-50s %8s %6s %9s %9s" % ("kernel", "wgs", "calls", "avg_us", "min_us"))
for (name, wg), v in sorted(agg.items(), key=lambda kv: (-tot[kv[0][0]], kv[0][0], -kv[0][1])):
    print("%-50s %8d %6d %9.1f %9.1f" % (name, wg, len(v), sum(v) / len(v), min(v)))
PY
head -70 $OUT/summary.txt
