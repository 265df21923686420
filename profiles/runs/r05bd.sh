cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05bd; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_sorblock_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
for i in 1 2 3; do
  for o in "sor_block_lean=xy" "sor_block_lean=hx"; do
    timeout 300 python3 profiles/micro/sb_ab.py d72 d96 128 plume d160 256 $o sor_block_lb=2 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
  done
done
sort -s -k5,5 $O/ab.txt
