cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05d; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 120 profiles/micro/valu_rate/valu_rate > $O/valu_rate.txt 2>&1
timeout 300 python3 bench.py > $O/bench256.json 2> $O/bench256.err
tail -c 3000 $O/valu_rate.txt
