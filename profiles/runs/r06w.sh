# round 6: final candidate (box2: both samples of advect_vector out of the box per wave; advect_scalars float-only form as committed; q4 form with the first sample out of the boxes too, six waves per SIMD)
# against the committed sources (prev)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06w; mkdir -p $O; rm -f $O/*
for rep in 1 2; do for l in prev box2; do
	HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$l.so timeout 300 python profiles/micro/advect_two_libs.py 256 128 plume1024 --amp=96,400 >> $O/ab.txt 2>>$O/err.txt
	HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$l.so timeout 300 python profiles/micro/bench_with_options.py - --no-cpu-baseline --no-strong 2>>$O/err.txt | sed "s/^/$l /" >> $O/ab.txt
	HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$l.so timeout 300 python profiles/micro/full_stage_series.py 256 40 1 > $O/series_${l}_$rep.json 2>>$O/err.txt
	HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$l.so timeout 300 python bench.py --full > $O/full_${l}_$rep.json 2>>$O/err.txt
	HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$l.so timeout 300 python bench.py --full --config 128 > $O/full128_${l}_$rep.json 2>>$O/err.txt
done; done
cat $O/ab.txt
python - <<'PY'
import json
L=("prev","box2")
def S(l,r):
    a=json.load(open(f"gpurun_out/r06w/series_{l}_{r}.json")); k=[x for x in a if x.startswith("us per")][0]; return a[k]
for r in (1,2):
    for lo,hi in ((4,24),(30,40)):
        for col,name in ((0,"advect_vector"),(4,"advect_scalars S=5")):
            print("rep",r,"substeps",lo,"-",hi-1,name,{l:round(sum(x[col] for x in S(l,r)[lo:hi])/(hi-lo),1) for l in L})
    print("rep",r,"bench --full",{l:round(json.loads(open(f"gpurun_out/r06w/full_{l}_{r}.json").read().strip().splitlines()[-1])["value"],1) for l in L}, "128:", {l:round(json.loads(open(f"gpurun_out/r06w/full128_{l}_{r}.json").read().strip().splitlines()[-1])["value"],1) for l in L})
PY
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_kernel_variants_gpu.py tests/test_operators_gpu.py tests/test_ref_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_dist_gpu.py -x -q > $O/pytest.log 2>&1
grep -n "passed\|failed" $O/pytest.log | tail -2
