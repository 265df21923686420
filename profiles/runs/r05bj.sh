cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05bj; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
B=$GRAFT_REPO_ROOT/profiles/micro/exp/libhns_base.so
timeout 900 python3 -m pytest tests/test_ref_kernels_gpu.py tests/test_parity_gpu.py tests/test_kernel_variants_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for c in 256 plume1024 512; do
  for i in 1 2; do
    echo "base $c $(HNS_LIBRARY=$B timeout 300 python3 profiles/micro/div_ab.py rev 1 1 $c 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-170)" >> $O/ab.txt
    echo "new  $c $(timeout 300 python3 profiles/micro/div_ab.py rev 1 1 $c 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-170)" >> $O/ab.txt
  done
done
cat $O/ab.txt
