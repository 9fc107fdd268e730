#!/bin/bash
# SQ-side counters of the two compositing kernels (own run, kernel dispatch only).
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS \
  --kernel-include-regex "render_(fwd|bwd)_kernel" -d $OUT/sq1 -o sq1 --output-format csv -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU \
  --kernel-include-regex "render_(fwd|bwd)_kernel" -d $OUT/sq2 -o sq2 --output-format csv -- $CMD > $OUT/sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT \
  --kernel-include-regex "render_(fwd|bwd)_kernel" -d $OUT/grbm -o grbm --output-format csv -- $CMD > $OUT/grbm.log 2>&1
python - <<PY
import csv,collections,glob
for f in sorted(glob.glob("$OUT/*/*_counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r['Kernel_Name'][:17],r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print(k, len(v), sum(v)/len(v))
PY
tail -2 $OUT/sq1.log $OUT/sq2.log
