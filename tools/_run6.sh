for e in "E3DGS_OVERLAP=0" "E3DGS_OVERLAP=1"; do
OUT=$PWD/gpurun_out/kt_$e; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && TMPDIR=/tmp env $e rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-substep > $OUT/log.txt 2>&1)
echo "== $e"; python tools/iter_kernels.py $OUT | grep -i "preprocess\|colour\|kernels "
done
bash tools/ab_base.sh 3 --steps 20 --warmup 5
