# round 6: the q4 advection at six waves per SIMD (80 registers, 28 spilled) against four (108 registers), after its forward taps moved to LDS
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06u; mkdir -p $O; rm -f $O/*
for rep in 1 2; do for l in prev w6; do
	HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$l.so timeout 300 python profiles/micro/full_stage_series.py 256 40 1 > $O/series_${l}_$rep.json 2>>$O/err.txt
	HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$l.so timeout 300 python bench.py --full > $O/full_${l}_$rep.json 2>>$O/err.txt
done; done
python - <<'PY'
import json
def S(l,r):
    a=json.load(open(f"gpurun_out/r06u/series_{l}_{r}.json")); k=[x for x in a if x.startswith("us per")][0]; return a[k]
for r in (1,2):
    for lo,hi in ((4,24),(30,40)):
        print("rep",r,"substeps",lo,"-",hi-1,"advect_scalars S=5",{l:round(sum(x[4] for x in S(l,r)[lo:hi])/(hi-lo),1) for l in ("prev","w6")})
    for l in ("prev","w6"):
        j=json.loads(open(f"gpurun_out/r06u/full_{l}_{r}.json").read().strip().splitlines()[-1]); print("rep",r,l,"bench --full substeps/s",round(j["value"],1))
PY
