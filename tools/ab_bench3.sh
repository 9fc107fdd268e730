#!/bin/bash
# As ab_bench.sh with any number of builds:  bash tools/ab_bench3.sh <rounds> <lib1.so> <lib2.so> ...
R=$1; shift
LIVE=event_3dgs_amd/libe3dgs_hip.so
cp $LIVE /tmp/live.so
for r in $(seq $R); do
  for src in "$@"; do
    cp $src $LIVE
    python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$src', d['value'],d['ms_per_step'], 'render_bwd', d['stages']['render_bwd']['avg_ms'])"
  done
done
cp /tmp/live.so $LIVE
