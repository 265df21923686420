cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05o; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests/test_ref_kernels_gpu.py tests/test_parity_gpu.py tests/test_kats.py tests/test_golden_kernels.py tests/test_operators_gpu.py tests/test_kernel_variants_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -2
for rep in 1 2 3; do
HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_r04.so timeout 300 python3 bench.py --no-cpu-baseline --no-strong 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('r04', round(d['value'],1), {n: round(1e3*v['ms_per_launch'],1) for n,v in k.items()})"
timeout 300 python3 bench.py --no-cpu-baseline --no-strong 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('new', round(d['value'],1), {n: round(1e3*v['ms_per_launch'],1) for n,v in k.items()})"
done
