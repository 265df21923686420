cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03v; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o lone -- python3 profiles/micro/dist_profile.py rank 1 0 > $O/log.txt 2>&1
grep "ms per substep" $O/log.txt
f=$(find $O/trace -name '*kernel_stats.csv' | head -1)
head -8 $f | cut -c1-220
t=$(find $O/trace -name '*kernel_trace.csv' | head -1)
n=$(wc -l < $t); python3 profiles/micro/trace_timeline.py $t $((n/2)) 75 > $O/timeline.txt
find $O/trace -name '*.csv' -size +20M -delete
