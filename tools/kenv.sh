#!/bin/bash
# Per-kernel A/B of ENVIRONMENT settings on the static workload (same library):
#   bash tools/kenv.sh "<kernel name regex>" "<N list>" "VAR=a" "VAR=b" ...      ("" = default environment)
# (tools/static_iter.py is the workload; two alternating rounds under rocprofv3 --kernel-trace --stats)
RE=$1; NS=$2; shift 2
REPO=$(pwd)
for N in $NS; do
for round in 1 2; do
for e in "$@"; do
  OUT=$REPO/gpurun_out/kenv_$(echo "$e" | tr -c 'A-Za-z0-9_=' '_')_$N; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && env $e TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python $REPO/tools/static_iter.py 12 $N > $OUT/log.txt 2>&1)
  python - <<PY
import csv,glob,re
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"].split("(")[0]
    if re.search(r"$RE", n): print("%-28s N=%-8s %-52s calls=%4s avg_us=%8.1f tot_ms=%8.2f" % ("[$e]", "$N", n[:52], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done; done; done
