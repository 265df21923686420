cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05r; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
for cfg in 256 128 plume plume1024 512; do
timeout 300 python3 profiles/micro/div_ab.py divergence row coalesced $cfg 2>&1 | grep -v amdgpu | tee -a $O/div_ab.txt
HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_advold.so timeout 300 python3 profiles/micro/div_ab.py divergence row coalesced $cfg 2>&1 | grep -v amdgpu | sed 's/^/before: /' | tee -a $O/div_ab.txt
done
timeout 600 python3 -m pytest tests/test_ref_kernels_gpu.py tests/test_kernel_variants_gpu.py -x -q -m gpu -k "diverg or variant or Compute or compute" 2>&1 | tail -2
