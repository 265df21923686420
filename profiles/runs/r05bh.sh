cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05bh; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_dist_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $GRAFT_REPO_ROOT/profiles/micro/dist_overhead.py plume1024 8 2 --partition > $O/run.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -rf $O/stats
grep "k_rbgs_block_xy<false, hns::PhaseMirror" $O/kernel_stats.csv | cut -c1-50,170-260
cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 600 python3 profiles/micro/dist_overhead.py plume1024 8 2 --partition --rank=4 2>&1 | grep -v amdgpu.ids | grep config >> $O/overhead.jsonl; done
python3 - <<'PY'
import json
for l in open("/root/repo/gpurun_out/r05bh/overhead.jsonl"):
    j = json.loads(l); print(j["all_ranks_lockstep_ms"], j["one_rank_loopback"]["substep_ms"], j["one_rank_loopback"]["pressure_us_per_iteration"])
PY
