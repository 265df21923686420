cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05i; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
HNS_SB_PERSIST=0 timeout 300 python3 profiles/micro/sb_ab.py 256 512 2>&1 | grep -v amdgpu.ids | sed "s/^/persist0 /" >> $O/ab.txt
for sg in 0 3 5 7 10; do
HNS_SB_PERSIST=1 HNS_SB_STAGGER=$sg timeout 300 python3 profiles/micro/sb_ab.py 256 512 2>&1 | grep -v amdgpu.ids | sed "s/^/persist1 stagger$sg /" >> $O/ab.txt
done
for sg in 0 7; do
HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_prio.so HNS_SB_PERSIST=1 HNS_SB_STAGGER=$sg timeout 300 python3 profiles/micro/sb_ab.py 256 512 2>&1 | grep -v amdgpu.ids | sed "s/^/persist1 prio2 stagger$sg /" >> $O/ab.txt
done
HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_prio.so HNS_SB_PERSIST=0 timeout 300 python3 profiles/micro/sb_ab.py 256 512 2>&1 | grep -v amdgpu.ids | sed "s/^/persist0 prio2 /" >> $O/ab.txt
done
cat $O/ab.txt
