# round 6: small ranks in line (pressure sweeps, divergence, gradient as one launch over the owned leaves + exchange on the compute stream): dist tests incl. the new real-data
# multi-process cases, then rank 4 / 0 / 7 of 8 of config 5 alone, fresh fields per repetition
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06f; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_dist_gpu.py -x -q 2>&1 | tail -8 > $O/pytest.log
for us in 1 0; do for k in 0 2; do for tr in "" "--rccl"; do
timeout 300 python3 profiles/micro/dist_exchanged_one.py $k $tr --three dist_unsplit=$us 2>&1 | grep "loopback_substep" | sed "s/^/dist_unsplit $us k $k /" | cut -c1-520 >> $O/t.txt
done; done; done
for r in 0 2 7; do timeout 300 python3 profiles/micro/dist_exchanged_one.py 0 --three --rank=$r 2>&1 | grep "loopback_substep\|plain" | sed "s/^/rank$r k 0 /" | cut -c1-520 >> $O/t.txt; done
timeout 300 python3 profiles/micro/dist_overhead.py plume1024 8 0 --partition --rank=4 --lone-only --three 2>&1 | grep "loopback_substep\|plain" | sed "s/^/chained (dist_mirror default) /" | cut -c1-520 >> $O/t.txt
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/trace_k0 -- python3 $GRAFT_REPO_ROOT/profiles/micro/dist_exchanged_one.py 0 > $O/trace_k0.log 2>&1
for f in $(find $O -name "*kernel_trace.csv" -o -name "*memory_copy_trace.csv"); do (head -1 $f; tail -5000 $f) > $f.tail; rm $f; done
find $O -name "*agent_info*" -delete
cat $O/pytest.log; cat $O/t.txt
