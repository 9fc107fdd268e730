#!/bin/bash
# usage: static_kcmp.sh lib1 lib2 ... : per-kernel durations of tools/static_iter.py (fixed parameters, no optimizer) for each build
REPO=$(pwd)
cp event_3dgs_amd/libe3dgs_hip.so /tmp/live.so
for round in 1 2; do
for lib in "$@"; do
  cp $lib event_3dgs_amd/libe3dgs_hip.so
  OUT=$REPO/gpurun_out/skcmp_$(basename $lib .so); rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python $REPO/tools/static_iter.py 40 > $OUT/log.txt 2>&1)
  python - <<PY
import csv,glob,collections
f=glob.glob("$OUT/**/*kernel_trace.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"].split("(")[0][:60]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("== $lib", open("$OUT/log.txt").read().strip().splitlines()[-1])
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:8]:
    v=sorted(v); top=v[len(v)//2:]
    print("%-62s n=%3d upper-half avg %8.1f us  min %8.1f" % (k, len(v), sum(top)/len(top), v[0]))
PY
done
done
cp /tmp/live.so event_3dgs_amd/libe3dgs_hip.so
