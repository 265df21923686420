cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04v; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o lone -- python3 profiles/micro/dist_lone.py plume1024 8 4 1 --partition > $O/log.txt 2>&1
grep "ms per substep" $O/log.txt | cut -c1-80
f=$(find $O/trace -name '*kernel_stats.csv' | head -1)
head -12 $f | cut -c1-150
t=$(find $O/trace -name '*kernel_trace.csv' | head -1)
n=$(wc -l < $t); python3 profiles/micro/trace_timeline.py $t $((n/2)) 70 > $O/timeline.txt
find $O/trace -name '*.csv' -size +20M -delete
