cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05be; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
E=$GRAFT_REPO_ROOT/profiles/micro/exp/libhns_eager.so
for i in 1 2 3; do
  timeout 300 python3 profiles/micro/sb_ab.py d72 128 plume 256 512 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
  HNS_LIBRARY=$E timeout 300 python3 profiles/micro/sb_ab.py d72 128 plume 256 512 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
done
sort -s -k1,1 -k3,3 $O/ab.txt
