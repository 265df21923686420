cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05g; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_sorblock_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for rep in 1 2; do
HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_r04.so timeout 300 python3 profiles/micro/sb_ab.py 128 256 plume1024 512 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
HNS_SB_PERSIST=1 timeout 300 python3 profiles/micro/sb_ab.py 128 256 plume1024 512 2>&1 | grep -v amdgpu.ids | sed 's/^/persist1 /' >> $O/ab.txt
HNS_SB_PERSIST=0 timeout 300 python3 profiles/micro/sb_ab.py 128 256 plume1024 512 2>&1 | grep -v amdgpu.ids | sed 's/^/persist0 /' >> $O/ab.txt
done
cat $O/ab.txt
