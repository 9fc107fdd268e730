#!/bin/bash
# SQ-side counters of the two compositing kernels (own runs, kernel dispatch only).  Usage: bash profiles/run_pmc_sq.sh <tag>
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/kbench.py --iters 4"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS \
  --kernel-include-regex "render_(fwd|bwd)_kernel" -d $OUT/sq1 -o sq1 --output-format csv -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU \
  --kernel-include-regex "render_(fwd|bwd)_kernel" -d $OUT/sq2 -o sq2 --output-format csv -- $CMD > $OUT/sq2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU \
  --kernel-include-regex "render_(fwd|bwd)_kernel" -d $OUT/sq3 -o sq3 --output-format csv -- $CMD > $OUT/sq3.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-include-regex "render_(fwd|bwd)_kernel" -d $OUT/grbm -o grbm --output-format csv -- $CMD > $OUT/grbm.log 2>&1
python - <<PY
import csv,collections,glob
for f in sorted(glob.glob("$OUT/*/*_counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r['Kernel_Name'][:17],r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print(k[0], k[1], len(v), round(sum(v)/len(v)))
PY
grep -l "rror" $OUT/*.log | head
