#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (KB) of EVERY kernel of the training step, max over launches (= the 3-view launches of the
# timed steps), two separate passes.  Usage (GPU box, repo root): bash tools/pmc_all_traffic.sh
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_all; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-substep"
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write --output-format csv -- $CMD > $OUT/write.log 2>&1
python - <<PY
import csv,collections,glob
agg=collections.defaultdict(dict)
SHARED=("radix_hist_kernel","radix_scatter_kernel","void scan_chained_kernel<false>")   # depth sort AND tile sort use them
for f in sorted(glob.glob("$OUT/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:40]; c=r['Counter_Name']; v=float(r['Counter_Value'])
        agg[k][c]=max(agg[k].get(c,0.0),v)
        if k in SHARED:                      # also per launch size: "<kernel> @<workgroups>"
            kk="%s @%d" % (k, int(r['Grid_Size'])//max(int(r.get('Workgroup_Size',256) or 256),1))
            agg[kk][c]=max(agg[kk].get(c,0.0),v)
print("%-42s %12s %12s   (MB per launch, max over launches; FETCH_SIZE x2 = gfx950 correction of the guide)" % ("kernel","fetch_MB","write_MB"))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1].get('FETCH_SIZE',0)-kv[1].get('WRITE_SIZE',0)):
    print("%-42s %12.1f %12.1f" % (k, 2*v.get('FETCH_SIZE',0)/1024, v.get('WRITE_SIZE',0)/1024))
PY
