cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05c; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
for c in "256 2 0 1" "plume1024 8 4 1 --partition"; do
n=$(echo $c | cut -d' ' -f1)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$n -o lone -- python3 profiles/micro/dist_lone.py $c > $O/log_$n.txt 2>&1
grep "ms per substep" $O/log_$n.txt | cut -c1-70
f=$(find $O/trace_$n -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats_$n.csv
t=$(find $O/trace_$n -name '*kernel_trace.csv' | head -1)
r=$(wc -l < $t); python3 profiles/micro/trace_timeline.py $t $((r/2)) 120 > $O/timeline_$n.txt
rm -rf $O/trace_$n
done
