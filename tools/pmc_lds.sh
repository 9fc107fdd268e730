REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_lds; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES --kernel-include-regex "render_bwd_kernel" -d $OUT/a -o a --output-format csv -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-substep > $OUT/a.log 2>&1
python - <<PY
import csv,collections,glob
for f in sorted(glob.glob("$OUT/*/*_counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r['Kernel_Name'][:24],r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print(k[0], k[1], len(v), "max", round(max(v)))
PY
