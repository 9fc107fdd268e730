#!/bin/bash
# usage: kcmp.sh lib1 lib2 ... : per-kernel avg durations (3-view launches: max class) under rocprofv3 for each library
REPO=$(pwd)
cp event_3dgs_amd/libe3dgs_hip.so /tmp/live.so
for lib in "$@"; do
  cp $lib event_3dgs_amd/libe3dgs_hip.so
  OUT=$REPO/gpurun_out/kcmp_$(basename $lib .so); rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-substep > $OUT/log.txt 2>&1)
  python - <<PY
import csv,glob,collections
f=glob.glob("$OUT/**/*kernel_trace.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"].split("(")[0][:60]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("== $lib")
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:22]:
    v=sorted(v); top=v[len(v)//2:]   # upper half = the 3-view launches
    print("%-62s n=%3d upper-half avg %8.1f us  min %8.1f" % (k, len(v), sum(top)/len(top), v[0]))
PY
done
cp /tmp/live.so event_3dgs_amd/libe3dgs_hip.so
