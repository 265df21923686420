cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05bm; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 1500 python3 tests/manual/stress_mirror_processes.py 3 > $O/stress.txt 2>&1; tail -16 $O/stress.txt
