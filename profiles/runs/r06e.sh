# round 6: the exchanged pressure loop, unsplit (one launch over all owned leaves packs its messages; transfer and unpack behind it on the compute stream) vs split + pipelined:
# the dist GPU tests, then rank 4 of 8 of config 5 alone (loopback: copies and RCCL), sweeps_per_exchange 4 (default) / 2, dist_pipeline 1 / 0
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06e; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_dist_gpu.py -x -q 2>&1 | tail -8 > $O/pytest.log
for pl in 1 0; do for k in 0 2; do for tr in "" "--rccl"; do
timeout 300 python3 profiles/micro/dist_exchanged_one.py $k $tr --three dist_unsplit=$pl 2>&1 | grep "loopback_substep" | sed "s/^/dist_unsplit $pl k $k /" | cut -c1-520 >> $O/t.txt
done; done; done
timeout 300 python3 profiles/micro/dist_exchanged_one.py 2 --three --rank=0 2>&1 | grep "loopback_substep" | sed "s/^/rank0 k 2 /" | cut -c1-520 >> $O/t.txt
timeout 300 python3 profiles/micro/dist_exchanged_one.py 2 --three --rank=7 2>&1 | grep "loopback_substep" | sed "s/^/rank7 k 2 /" | cut -c1-520 >> $O/t.txt
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/trace_k2 -- python3 $GRAFT_REPO_ROOT/profiles/micro/dist_exchanged_one.py 2 > $O/trace_k2.log 2>&1
for f in $(find $O -name "*kernel_trace.csv" -o -name "*memory_copy_trace.csv"); do (head -1 $f; tail -5000 $f) > $f.tail; rm $f; done
find $O -name "*agent_info*" -delete
cat $O/pytest.log; cat $O/t.txt
