cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05ba; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 600 python3 profiles/micro/dist_overhead.py plume1024 8 2 --partition --rank=4 2>&1 | grep -v amdgpu.ids | grep config >> $O/overhead.jsonl; done
python3 - <<'PY'
import json
for l in open("/root/repo/gpurun_out/r05ba/overhead.jsonl"):
    j = json.loads(l); print(j["all_ranks_lockstep_ms"], j["one_rank_loopback"]["substep_ms"], j["one_rank_loopback"]["pressure_us_per_iteration"])
PY
timeout 900 python3 -m pytest tests/test_dist_gpu.py -x -q -m gpu -k "chained or local_ranks or processes" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
