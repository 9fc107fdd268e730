set -x
python -m pytest tests/test_hip_train.py -x -q -k "adam or fused_step or apply_update" 2>&1 | tail -3
bash tools/ab_env.sh 3 "--steps 20 --warmup 5" "E3DGS_ADAM_GAP=0" "E3DGS_ADAM_GAP=1" 2>&1 | grep "^\["
for c in cfg2_200k_800px cfg3_1M_1080p_event; do
OUT=$PWD/gpurun_out/kt_$c; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 12 --warmup 3 --no-cpu-baseline --no-substep > $OUT/log.txt 2>&1)
python tools/iter_kernels.py $OUT > gpurun_out/iter_$c.txt 2>&1
tail -3 gpurun_out/iter_$c.txt
done
