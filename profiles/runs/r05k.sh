cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05k; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
HNS_SB_ORDER=1 timeout 900 python3 -m pytest tests/test_sorblock_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for rep in 1 2; do
for od in 0 1 2; do
HNS_SB_ORDER=$od timeout 300 python3 profiles/micro/sb_ab.py 128 256 plume1024 512 2>&1 | grep -v amdgpu.ids | sed "s/^/order$od /" >> $O/ab.txt
done
for pr in prio2 prio3; do
HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$pr.so timeout 300 python3 profiles/micro/sb_ab.py 128 256 plume1024 512 2>&1 | grep -v amdgpu.ids | sed "s/^/order0 /" >> $O/ab.txt
done
HNS_SB_ORDER=1 HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_prio2.so timeout 300 python3 profiles/micro/sb_ab.py 128 256 plume1024 512 2>&1 | grep -v amdgpu.ids | sed "s/^/order1 /" >> $O/ab.txt
done
cat $O/ab.txt
