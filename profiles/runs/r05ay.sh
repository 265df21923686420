cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05ay; mkdir -p $O; rm -rf $O/*
for r in 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$r -- python3 $GRAFT_REPO_ROOT/profiles/micro/dist_overhead.py plume1024 8 2 --partition --rank=$r --lone-only > $O/lone_$r.log 2>&1
  cp $(find $O/stats_$r -name "*kernel_stats.csv" | head -1) $O/kernel_stats_rank$r.csv; rm -rf $O/stats_$r
  grep -v amdgpu.ids $O/lone_$r.log | tail -3; head -14 $O/kernel_stats_rank$r.csv | cut -c1-200
done
