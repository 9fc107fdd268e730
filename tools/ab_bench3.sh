#!/bin/bash
# A/B of any number of library builds on the SAME GPU box:  bash tools/ab_bench3.sh <rounds> <lib1.so> <lib2.so> ...
R=$1; shift
LIVE=event_3dgs_amd/libe3dgs_hip.so
cp $LIVE /tmp/live.so
for r in $(seq $R); do
  for src in "$@"; do
    cp $src $LIVE
    python bench.py --no-cpu-baseline --no-substep $BENCH_ARGS 2>&1 | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$src', d['value'],d['ms_per_step'], {k:v['avg_ms'] for k,v in d['stages'].items()})"
  done
done
cp /tmp/live.so $LIVE
