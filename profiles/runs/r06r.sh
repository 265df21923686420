# round 6: the 10^3 box decided per WAVE (libhns_boxwave.so) against per LANE (libhns_boxlane.so) against the sources before (libhns_prev.so): the full substep's stage series
# (transient substeps 4-23, settled 30-39), bench.py --full and the core bench, alternating in one call
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06r; mkdir -p $O; rm -f $O/*
for rep in 1 2; do for l in prev boxwave boxlane; do
	HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$l.so timeout 300 python profiles/micro/full_stage_series.py 256 40 1 > $O/series_${l}_$rep.json 2>>$O/err.txt
	HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$l.so timeout 300 python bench.py --full > $O/full_${l}_$rep.json 2>>$O/err.txt
	HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$l.so timeout 300 python profiles/micro/bench_with_options.py - --no-cpu-baseline --no-strong 2>>$O/err.txt | sed "s/^/$l /"
done; done
python - <<'PY'
import json
def S(l,r):
    a=json.load(open(f"gpurun_out/r06r/series_{l}_{r}.json")); k=[x for x in a if x.startswith("us per")][0]; return a[k]
for r in (1,2):
    for lo,hi in ((4,24),(9,20),(30,40)):
        for col,name in ((0,"advect_vector"),(4,"advect_scalars"),(2,"pressure")):
            print("rep",r,"substeps",lo,"-",hi-1,name,{l:round(sum(x[col] for x in S(l,r)[lo:hi])/(hi-lo),1) for l in ("prev","boxwave","boxlane")})
    for l in ("prev","boxwave","boxlane"):
        j=json.loads(open(f"gpurun_out/r06r/full_{l}_{r}.json").read().strip().splitlines()[-1]); print("rep",r,l,"bench --full substeps/s",round(j["value"],1))
PY
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_kernel_variants_gpu.py tests/test_operators_gpu.py tests/test_ref_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_dist_gpu.py -x -q > $O/pytest.log 2>&1
grep -n "passed\|failed" $O/pytest.log | tail -2
