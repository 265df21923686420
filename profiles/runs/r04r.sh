cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04r; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o lone -- python3 profiles/micro/dist_lone.py 256 2 0 1 > $O/log.txt 2>&1
grep "ms per substep" $O/log.txt | cut -c1-80
f=$(find $O/trace -name '*kernel_stats.csv' | head -1)
head -9 $f | cut -c1-170
t=$(find $O/trace -name '*kernel_trace.csv' | head -1)
n=$(wc -l < $t); python3 profiles/micro/trace_timeline.py $t $((n/2)) 70 > $O/timeline.txt
find $O/trace -name '*.csv' -size +20M -delete
