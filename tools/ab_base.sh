#!/bin/bash
# Same-box A/B of the working tree against the round's baseline tree:  bash tools/ab_base.sh <rounds> [bench args]
# (.base/ = `git archive <baseline commit> | tar -x -C .base`, built with its own __graft_entry__.build(); git-ignored)
R=${1:-2}; shift
for r in $(seq $R); do
  for d in .base .; do
    (cd $d && python bench.py --no-cpu-baseline --no-substep "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[$d]', d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['stages'].items()})")
  done
done
