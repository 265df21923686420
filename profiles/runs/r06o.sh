# round 6: far taps, hash walks (libhns_hash.so = the committed sources) against two hops through the neighbour tables (libhns_twohop.so), alternating in one call: the stage series of the
# full substep at 256^3 (40 substeps behind an upload) and bench.py --full
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06o; mkdir -p $O; rm -f $O/*
for rep in 1 2; do for l in hash twohop; do
	HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$l.so timeout 300 python profiles/micro/full_stage_series.py 256 40 1 > $O/series_${l}_$rep.json 2>>$O/err.txt
	HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$l.so timeout 300 python bench.py --full > $O/full_${l}_$rep.json 2>>$O/err.txt
done; done
python - <<'PY'
import json
def S(l,r):
    a=json.load(open(f"gpurun_out/r06o/series_{l}_{r}.json")); k=[x for x in a if x.startswith("us per")][0]; return a[k]
for r in (1,2):
    h,t=S("hash",r),S("twohop",r)
    for lo,hi in ((4,24),(9,20),(30,40)):
        for col,name in ((0,"advect_vector"),(4,"advect_scalars")):
            print("rep",r,"substeps",lo,"-",hi-1,name,"hash",round(sum(x[col] for x in h[lo:hi])/(hi-lo),1),"two hops",round(sum(x[col] for x in t[lo:hi])/(hi-lo),1))
    for l in ("hash","twohop"):
        j=json.loads(open(f"gpurun_out/r06o/full_{l}_{r}.json").read().strip().splitlines()[-1]); print("rep",r,l,"bench --full substeps/s",round(j["value"],1))
PY
