# round 6: the 10^3 LDS box in advect_vector AND advect_scalars (S = 1 and the q4 form): libhns_box.so against libhns_prev.so, alternating; parity tests; bench lines (core and --full)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06q; mkdir -p $O; rm -f $O/*
for rep in 1 2; do for l in prev box; do
	HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$l.so timeout 300 python profiles/micro/advect_two_libs.py 256 128 plume1024 --amp=96,160,400 >> $O/ab.txt 2>>$O/err.txt
done; done
cat $O/ab.txt
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_kernel_variants_gpu.py tests/test_operators_gpu.py tests/test_ref_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_dist_gpu.py -x -q > $O/pytest.log 2>&1
grep -n "passed\|failed" $O/pytest.log | tail -2
for l in prev box prev box; do HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$l.so timeout 300 python profiles/micro/bench_with_options.py - --no-cpu-baseline --no-strong | sed "s/^/$l /"; done
for l in prev box prev box; do HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$l.so timeout 300 python bench.py --full > $O/full_$l.json; python -c "
import json; j=json.loads(open('$O/full_$l.json').read().strip().splitlines()[-1]); print('$l --full', round(j['value'],1), {k:round(v['ms_per_substep']*1000) for k,v in j['roofline']['kernels'].items()})"; done
