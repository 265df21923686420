# round 6: BFECC's second sample out of a 10^3 LDS box (libhns_box.so) against the sources before (libhns_prev.so): kernels alternating in one call, then parity tests, then bench lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p; mkdir -p $O; rm -f $O/*
for rep in 1 2; do for l in prev box; do
	HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$l.so timeout 300 python profiles/micro/advect_two_libs.py 256 128 plume1024 --amp=96,160,400 >> $O/ab.txt 2>>$O/err.txt
done; done
cat $O/ab.txt
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_kernel_variants_gpu.py tests/test_operators_gpu.py tests/test_ref_kernels_gpu.py tests/test_fullsize_gpu.py -x -q > $O/pytest.log 2>&1
grep -n "passed\|failed" $O/pytest.log | tail -2
for l in prev box prev box; do HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$l.so timeout 300 python profiles/micro/bench_with_options.py - --no-cpu-baseline --no-strong | sed "s/^/$l /"; done
