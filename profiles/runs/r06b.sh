# round 6: kernel + copy traces of the exchanged pressure loop of rank 4 of 8 of config 5 (loopback), sweeps_per_exchange 4 (default) and 2, and over the RCCL loopback
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06b; mkdir -p $O; rm -rf $O/*
for k in 0 2; do
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/trace_k$k -- python3 $GRAFT_REPO_ROOT/profiles/micro/dist_exchanged_one.py $k > $O/trace_k$k.log 2>&1
done
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/trace_k2_rccl -- python3 $GRAFT_REPO_ROOT/profiles/micro/dist_exchanged_one.py 2 --rccl > $O/trace_k2_rccl.log 2>&1
for f in $(find $O -name "*kernel_trace.csv" -o -name "*memory_copy_trace.csv"); do (head -1 $f; tail -5000 $f) > $f.tail; rm $f; done
find $O -name "*agent_info*" -delete
du -sh $O; ls -R $O | head -30
