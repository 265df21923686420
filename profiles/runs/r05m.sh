cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05m; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_dist_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for k in 2 4; do
for rk in 4 6; do
timeout 300 python3 profiles/micro/dist_overhead.py plume1024 8 $k --partition --rank=$rk >> $O/overhead.jsonl 2>> $O/overhead.err
timeout 300 python3 profiles/micro/dist_overhead.py plume1024 8 $k --partition --rank=$rk --leaf-order >> $O/overhead.jsonl 2>> $O/overhead.err
done; done
cat $O/overhead.jsonl | cut -c1-900
timeout 600 python3 bench.py --gpus 2 --share-one-gpu --steps 5 > $O/bench_2procs.json 2> $O/bench_2procs.err; tail -c 3000 $O/bench_2procs.json; tail -5 $O/bench_2procs.err
