cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03s; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o loc -- python3 profiles/micro/dist_local.py 256 2 1 > $O/log.txt 2>&1
grep "ms per substep" $O/log.txt
f=$(find $O/trace -name '*kernel_stats.csv' | head -1)
head -12 $f | cut -c1-200
t=$(find $O/trace -name '*kernel_trace.csv' | head -1)
n=$(wc -l < $t); python3 profiles/micro/trace_timeline.py $t $((n/2)) 30 > $O/timeline.txt
find $O/trace -name '*.csv' -size +20M -delete
