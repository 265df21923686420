cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05bc; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
E=$GRAFT_REPO_ROOT/profiles/micro/exp/libhns_rreg.so
HNS_LIBRARY=$E timeout 900 python3 -m pytest tests/test_sorblock_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for i in 1 2 3; do
  timeout 300 python3 profiles/micro/sb_ab.py 128 256 plume1024 512 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
  HNS_LIBRARY=$E timeout 300 python3 profiles/micro/sb_ab.py 128 256 plume1024 512 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
done
cat $O/ab.txt
