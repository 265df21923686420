mkdir -p gpurun_out/r02c
for k in 4 2; do python profiles/micro/dist_overhead.py 256 2 $k >> gpurun_out/r02c/overhead.jsonl 2>> gpurun_out/r02c/err.log; done
python profiles/micro/dist_overhead.py 128 2 4 >> gpurun_out/r02c/overhead.jsonl 2>> gpurun_out/r02c/err.log
cat gpurun_out/r02c/overhead.jsonl
grep -v amdgpu.ids gpurun_out/r02c/err.log | tail -5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02c/prof -- python3 $GRAFT_REPO_ROOT/profiles/micro/dist_overhead.py 256 2 4 > $GRAFT_REPO_ROOT/gpurun_out/r02c/prof.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r02c/prof -name "*kernel_stats.csv" | head -1); head -20 $f | cut -c1-160
