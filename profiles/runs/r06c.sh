# round 6: the exchanged substep as a hipGraph: tests, then rank 4 of 8 of config 5 alone (loopback: copies and RCCL), dist_graph 1 / 0, sweeps_per_exchange 4 (default) / 2
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06c; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
HNS_DIST_GRAPH_VERBOSE=1 timeout 900 python3 -m pytest tests/test_dist_gpu.py -x -q -k "graph or loopback or rccl_carries" 2>&1 | tail -15 > $O/pytest.log
for g in 1 0; do for k in 0 2; do for tr in "" "--rccl"; do
HNS_DIST_GRAPH_VERBOSE=1 timeout 300 python3 profiles/micro/dist_exchanged_one.py $k $tr --three dist_graph=$g 2>&1 | grep -v amdgpu.ids | sed "s/^/dist_graph $g k $k /" | cut -c1-700 >> $O/t.txt
done; done; done
cat $O/pytest.log; cat $O/t.txt
