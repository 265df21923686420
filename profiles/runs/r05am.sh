cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05am; mkdir -p $O; rm -rf $O/*
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $GRAFT_REPO_ROOT/profiles/micro/dist_overhead.py plume1024 8 2 --partition > $O/run.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_lockstep.csv; rm -rf $O/stats
grep -v amdgpu.ids $O/run.log | grep config | cut -c1-600; head -24 $O/kernel_stats_lockstep.csv | cut -c1-230
