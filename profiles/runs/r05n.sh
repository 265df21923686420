cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05n; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for k in 2 4; do
for rk in 4 6; do
timeout 300 python3 profiles/micro/dist_overhead.py plume1024 8 $k --partition --rank=$rk >> $O/overhead.jsonl 2>> $O/overhead.err
done; done
python3 - <<'PY'
import json
for l in open("/root/repo/gpurun_out/r05n/overhead.jsonl"):
    d=json.loads(l); print(d["sweeps_per_exchange"], d["partition_axis"], d["all_ranks_lockstep_ms"], d["one_rank_loopback"])
PY
timeout 600 python3 bench.py --cook > $O/cook256.json 2> $O/cook256.err; cat $O/cook256.json
