cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05h; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
HNS_SB_PERSIST=2 timeout 900 python3 -m pytest tests/test_sorblock_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for rep in 1 2; do
for m in 0 1 2; do
HNS_SB_PERSIST=$m timeout 300 python3 profiles/micro/sb_ab.py 256 plume1024 512 2>&1 | grep -v amdgpu.ids | sed "s/^/persist$m /" >> $O/ab.txt
done; done
cat $O/ab.txt
