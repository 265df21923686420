cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05x; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 3000 python3 -m pytest tests/test_sorblock_gpu.py tests/test_kernel_variants_gpu.py tests/test_fullsize_gpu.py tests/test_ref_kernels_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -2
for c in plume1024 512; do timeout 300 python3 bench.py --config $c --no-cpu-baseline --steps 10 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$c', round(d['value'],1), 'SOR', round(1e3*r['ms_per_launch'],1), round(r['frac'],3), r['kernel'][:40])"; done
