cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05ag; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests/test_sorblock_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
for i in 1 2; do
  for r in 0 1; do
    timeout 300 python3 profiles/micro/sb_ab.py 128 plume d96 sor_resident=$r 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
  done
done
cat $O/ab.txt
