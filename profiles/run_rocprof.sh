#!/bin/bash
# Usage (on the GPU box, from the repo root):  bash profiles/run_rocprof.sh <tag>
# Produces gpurun_out/prof_<tag>/{trace,fetch,write}/... ; summaries are copied into profiles/ by hand.
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-substep"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- $CMD > $OUT/trace.log 2>&1
# PMC passes: own runs, no trace domains besides kernel dispatch (see task notes)
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "render_(fwd|bwd)_kernel" -d $OUT/fetch -o fetch --output-format csv -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "render_(fwd|bwd)_kernel" -d $OUT/write -o write --output-format csv -- $CMD > $OUT/write.log 2>&1
find $OUT -name "*.csv" | head -20
tail -3 $OUT/trace.log
