#!/bin/bash
# HBM traffic + SQ counters of ONE kernel (regex) inside the training step.  Own runs, kernel dispatch only.
# Usage (GPU box, repo root): bash profiles/run_pmc_kernel.sh <regex> <tag>
RE=${1:-geom_bwd_multi_kernel}
TAG=${2:-k}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmck_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-substep"
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$RE" -d $OUT/fetch -o fetch --output-format csv -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$RE" -d $OUT/write -o write --output-format csv -- $CMD > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
  --kernel-include-regex "$RE" -d $OUT/sq -o sq --output-format csv -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-include-regex "$RE" -d $OUT/tcc -o tcc --output-format csv -- $CMD > $OUT/tcc.log 2>&1
rocprofv3 --kernel-trace --stats --kernel-include-regex "$RE" -d $OUT/trace -o trace --output-format csv -- $CMD > $OUT/trace.log 2>&1
python - <<PY
import csv,collections,glob
for f in sorted(glob.glob("$OUT/*/*_counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r['Kernel_Name'][:24],r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print(k[0], k[1], len(v), round(sum(v)/len(v)))
for f in glob.glob("$OUT/trace/*kernel_stats.csv"):
    print(open(f).read()[:1500])
PY
grep -l "rror" $OUT/*.log | head
