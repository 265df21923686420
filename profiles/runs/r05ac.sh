cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05ac; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests/test_dist_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -2
timeout 1500 python3 tests/manual/stress_mirror_processes.py 3 > $O/stress.txt 2>&1; tail -14 $O/stress.txt
