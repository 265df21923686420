cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05at; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for o in "sor_block_lb=1 sor_block_k=4" "sor_block_lb=1 sor_block_k=2" "sor_block_lb=2"; do
    timeout 300 python3 profiles/micro/sb_ab.py d32 d48 64 d72 d80 d96 $o 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
  done
done
sort -s -k4,4 -k2,3 $O/ab.txt
