#!/bin/bash
# Instruction mix / LDS counters of ONE kernel (regex) inside the training step; prints the MAX over launches
# (the 3-view launches of the timed steps).  Usage: bash profiles/run_pmc_mix.sh <regex> <tag>
RE=${1:-bin_kernel}
TAG=${2:-mix}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmcm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-substep"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES \
  --kernel-include-regex "$RE" -d $OUT/a -o a --output-format csv -- $CMD > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES \
  --kernel-include-regex "$RE" -d $OUT/b -o b --output-format csv -- $CMD > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_TRANS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --kernel-include-regex "$RE" -d $OUT/c -o c --output-format csv -- $CMD > $OUT/c.log 2>&1
python - <<PY
import csv,collections,glob
for f in sorted(glob.glob("$OUT/*/*_counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r['Kernel_Name'][:24],r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print(k[0], k[1], len(v), "max", round(max(v)))
PY
grep -l "rror" $OUT/*.log | head
