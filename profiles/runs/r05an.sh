cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05an; mkdir -p $O; rm -rf $O/*
for l in nomst nobeg; do
export HNS_LIBRARY=$GRAFT_REPO_ROOT/profiles/micro/exp/libhns_$l.so
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $GRAFT_REPO_ROOT/profiles/micro/dist_overhead.py plume1024 8 2 --partition > $O/run_$l.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$l.csv; rm -rf $O/stats
echo $l; grep "PhaseMirror, false" $O/kernel_stats_$l.csv | cut -c1-60,150-260
done
