cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05aa; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests/test_dist_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -2
for k in 2 4; do for rk in 4 6; do
timeout 300 python3 profiles/micro/dist_overhead.py plume1024 8 $k --partition --rank=$rk >> $O/overhead.jsonl 2>> $O/overhead.err
done; done
python3 - <<'PY'
import json
for l in open("/root/repo/gpurun_out/r05aa/overhead.jsonl"):
    d=json.loads(l); print(d["sweeps_per_exchange"], d["partition_axis"], d["all_ranks_lockstep_ms"], d["lockstep_host_enqueue_ms_per_rank_per_substep"], d["one_rank_loopback"]["rank"], d["one_rank_loopback"]["substep_ms"], d["one_rank_loopback"]["pressure_us_per_iteration"], d["one_rank_loopback"]["host_enqueue_ms"])
PY
