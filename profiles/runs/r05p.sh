cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05p; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 120 profiles/micro/gather_width/gather_width > $O/gather_width.txt 2>&1; cat $O/gather_width.txt
timeout 2400 python3 -m pytest tests/test_dist_gpu.py tests/test_operators_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -2
timeout 600 python3 profiles/micro/advect_lib_ab.py $PWD/profiles/micro/exp/libhns_advold.so $PWD/hnanosolver_amd/lib/libhns.so 256 > $O/advect_ab_256.txt 2>&1; cat $O/advect_ab_256.txt
timeout 600 python3 profiles/micro/advect_lib_ab.py $PWD/profiles/micro/exp/libhns_advold.so $PWD/hnanosolver_amd/lib/libhns.so plume1024 > $O/advect_ab_plume1024.txt 2>&1; cat $O/advect_ab_plume1024.txt
timeout 300 python3 bench.py --full > $O/full256.json 2> $O/full256.err; python3 -c "
import json; d=json.load(open('$O/full256.json')); print(round(d['value'],1), round(d['ms_per_step'],3), {k: (round(v['ms_per_substep'],3), round(v['frac'],3)) for k,v in d['roofline']['kernels'].items()})"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_full -o full -- python3 bench.py --full --steps 10 > $O/full256_prof.json 2> $O/full256_prof.err
f=$(find $O/prof_full -name '*kernel_stats.csv' | head -1); cp $f $O/full256_kernel_stats.csv; rm -rf $O/prof_full; head -12 $O/full256_kernel_stats.csv | cut -c1-160
