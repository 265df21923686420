cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05j; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
HNS_SB_PERSIST=0 timeout 300 python3 profiles/micro/sb_ab.py 256 512 2>&1 | grep -v amdgpu.ids | sed "s/^/persist0 /" >> $O/ab.txt
HNS_SB_VERBOSE=1 HNS_SB_PERSIST=1 timeout 300 python3 profiles/micro/sb_ab.py 256 512 2>&1 | grep -v amdgpu.ids | sed "s/^/persist1 auto /" >> $O/ab.txt
for gr in 512 768 1024 1536 2048; do
HNS_SB_GRID=$gr HNS_SB_PERSIST=1 timeout 300 python3 profiles/micro/sb_ab.py 256 512 2>&1 | grep -v amdgpu.ids | sed "s/^/persist1 grid$gr /" >> $O/ab.txt
done
done
cat $O/ab.txt
