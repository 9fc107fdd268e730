#!/bin/bash
# A/B of environment settings on the SAME GPU box and the same library:
#   bash tools/ab_env.sh <rounds> "<bench args>" "VAR=a" "VAR=b" ...      (an empty string "" = default environment)
R=$1; shift
ARGS=$1; shift
for r in $(seq $R); do
  for e in "$@"; do
    env $e python bench.py --no-cpu-baseline --no-substep $ARGS 2>&1 | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('[$e]', d['value'],d['ms_per_step'], {k:v['avg_ms'] for k,v in d['stages'].items()})"
  done
done
