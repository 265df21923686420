cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05bo; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for sg in 0 2 4 6 8; do
    timeout 300 python3 profiles/micro/sb_ab.py 128 plume d112 sor_block_stagger=$sg 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
  done
done
sort -s -k4,4 -k3,3 $O/ab.txt
