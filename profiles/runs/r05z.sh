cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05z; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 3000 python3 -m pytest tests/test_sorblock_gpu.py tests/test_kernel_variants_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -2
for rep in 1 2; do timeout 300 python3 profiles/micro/sb_ab.py plume1024 512 2>&1 | grep -v amdgpu.ids; done
