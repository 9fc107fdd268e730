#!/bin/bash
# Per-kernel A/B of library builds on the static workload:  bash tools/kab.sh "<kernel name regex>" "<N list>" libA.so libB.so ...
# (tools/build_variant.sh makes the builds; tools/static_iter.py is the workload; two alternating rounds under rocprofv3)
RE=$1; NS=$2; shift 2
REPO=$(pwd)
cp event_3dgs_amd/libe3dgs_hip.so /tmp/live.so
for N in $NS; do
for round in 1 2; do
for lib in "$@"; do
  cp $lib event_3dgs_amd/libe3dgs_hip.so
  OUT=$REPO/gpurun_out/kab_$(basename $lib .so)_$N; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python $REPO/tools/static_iter.py 12 $N > $OUT/log.txt 2>&1)
  python - <<PY
import csv,glob,re
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"].split("(")[0]
    if re.search(r"$RE", n): print("%-28s N=%-8s %-44s calls=%3s avg_us=%8.1f" % ("$(basename $lib)", "$N", n[:44], r["Calls"], float(r["AverageNs"])/1e3))
PY
done; done; done
cp /tmp/live.so event_3dgs_amd/libe3dgs_hip.so
