cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05aj; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
X=$GRAFT_REPO_ROOT/profiles/micro/exp
HNS_LIBRARY=$X/libhns_res16.so timeout 600 python3 -m pytest tests/test_sorblock_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
HNS_LIBRARY=$X/libhns_res16t.so timeout 300 python3 profiles/micro/res_trace.py 128 2>&1 | grep -v amdgpu.ids >> $O/trace.txt
cat $O/trace.txt
for i in 1 2; do
  timeout 300 python3 profiles/micro/sb_ab.py 128 plume d96 sor_resident=0 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
  for l in res16 res17; do
    HNS_LIBRARY=$X/libhns_$l.so timeout 300 python3 profiles/micro/sb_ab.py 128 plume d96 sor_resident=1 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
  done
done
cat $O/ab.txt
