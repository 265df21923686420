cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05w; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_divdma.so timeout 900 python3 -m pytest tests/test_sorblock_gpu.py tests/test_ref_kernels_gpu.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2 3; do
timeout 300 python3 profiles/micro/sb_ab.py 128 256 plume1024 512 2>&1 | grep -v amdgpu.ids | sed "s/^/base   /" >> $O/ab.txt
HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_divdma.so timeout 300 python3 profiles/micro/sb_ab.py 128 256 plume1024 512 2>&1 | grep -v amdgpu.ids | sed "s/^/divdma /" >> $O/ab.txt
done
sort $O/ab.txt
