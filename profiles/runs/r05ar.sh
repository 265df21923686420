cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05ar; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
X=$GRAFT_REPO_ROOT/profiles/micro/exp
for i in 1 2 3; do
  timeout 300 python3 profiles/micro/sb_ab.py 128 256 512 sor_block_lean=xy 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
  for l in pad3 pad4 pad8 conv; do
    HNS_LIBRARY=$X/libhns_$l.so timeout 300 python3 profiles/micro/sb_ab.py 128 256 512 sor_block_lean=xy 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
  done
done
sort -s -k1,1 -k4,4 $O/ab.txt
