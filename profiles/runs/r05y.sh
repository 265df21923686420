cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05y; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_divkdma.so timeout 600 python3 -m pytest tests/test_ref_kernels_gpu.py tests/test_kats.py -x -q -m gpu -k "diverg or kat or Compute or compute" 2>&1 | tail -2
for cfg in 256 plume1024 512; do
timeout 300 python3 profiles/micro/div_ab.py divergence coalesced coalesced $cfg 2>&1 | grep -v amdgpu | sed 's/^/base /' | cut -c1-200
HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_divkdma.so timeout 300 python3 profiles/micro/div_ab.py divergence coalesced coalesced $cfg 2>&1 | grep -v amdgpu | sed 's/^/dma  /' | cut -c1-200
done
