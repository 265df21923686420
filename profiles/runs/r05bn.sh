cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05bn; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for sg in 0 32 128 512; do
    timeout 300 python3 profiles/micro/sb_ab.py 256 plume1024 512 sor_block_seg=$sg 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
  done
done
sort -s -k4,4 -k3,3 $O/ab.txt
