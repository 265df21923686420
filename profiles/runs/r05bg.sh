cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05bg; mkdir -p $O; rm -rf $O/*
for l in base nomst; do
[ $l = nomst ] && export HNS_LIBRARY=$GRAFT_REPO_ROOT/profiles/micro/exp/libhns_nomst.so
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $GRAFT_REPO_ROOT/profiles/micro/dist_overhead.py plume1024 8 2 --partition > $O/run_$l.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$l.csv; rm -rf $O/stats
echo $l; grep "k_rbgs_block_xy<false, hns::PhaseMirror" $O/kernel_stats_$l.csv | cut -c1-50,170-260
done
