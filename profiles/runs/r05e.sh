cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05e; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 120 profiles/micro/valu_rate/valu_rate2 > $O/valu_rate2.txt 2>&1
grep "SIMD 8" $O/valu_rate2.txt
