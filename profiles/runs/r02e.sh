mkdir -p gpurun_out/r02e
python -m pytest tests/test_dist_gpu.py -m gpu -q -x --durations=8 2>&1 | tail -25 > gpurun_out/r02e/tests.log
cat gpurun_out/r02e/tests.log | tail -14
cd /tmp && export TMPDIR=/tmp
for m in single rank; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02e/prof_$m -- python3 $GRAFT_REPO_ROOT/profiles/micro/dist_profile.py $m > $GRAFT_REPO_ROOT/gpurun_out/r02e/prof_$m.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/r02e/prof_$m.log
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r02e/prof_$m -name "*kernel_stats.csv" | head -1); head -12 $f | cut -d, -f1-4 | cut -c1-150
done
