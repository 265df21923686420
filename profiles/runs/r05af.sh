cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05af; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
X=$GRAFT_REPO_ROOT/profiles/micro/exp
for i in 1 2 3; do
  for l in earlyids2 touch768 touch256; do
    HNS_LIBRARY=$X/libhns_$l.so timeout 300 python3 profiles/micro/sb_ab.py 128 256 plume1024 512 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
  done
done
cat $O/ab.txt
