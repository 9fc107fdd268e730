#!/bin/bash
# A/B two builds of the library on the SAME GPU box (box-to-box spread is ~3 %).
# Usage (GPU box, repo root):  bash tools/ab_bench.sh <libA.so> <libB.so> [rounds]
# Build the variants beforehand, e.g.  git stash; build; cp event_3dgs_amd/libe3dgs_hip.so event_3dgs_amd/lib_A.so
A=$1; B=$2; R=${3:-2}
LIVE=event_3dgs_amd/libe3dgs_hip.so
cp $LIVE /tmp/live.so
for r in $(seq $R); do
  for v in A B; do
    eval src=\$$v
    cp $src $LIVE
    python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v', d['value'],d['ms_per_step']);print('   ', {k:v['avg_ms'] for k,v in d['stages'].items()})"
  done
done
cp /tmp/live.so $LIVE
