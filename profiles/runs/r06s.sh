# round 6: stage series of the full substep and bench lines on the final build (10^3 box: per wave in advect_vector, per lane in advect_scalars)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06s; mkdir -p $O; rm -f $O/*
timeout 300 python profiles/micro/full_stage_series.py 256 40 1 > $O/series.json 2>>$O/err.txt
for i in 1 2 3; do timeout 300 python bench.py --full > $O/full_$i.json 2>>$O/err.txt; timeout 300 python bench.py --no-cpu-baseline --no-strong > $O/bench_$i.json 2>>$O/err.txt; done
python - <<'PY'
import json
a=json.load(open("gpurun_out/r06s/series.json")); k=[x for x in a if x.startswith("us per")][0]; r=a[k]
for lo,hi in ((4,24),(9,20),(30,40)): print("substeps",lo,"-",hi-1,[round(sum(x[c] for x in r[lo:hi])/(hi-lo),1) for c in range(5)])
for i in (1,2,3):
    j=json.loads(open(f"gpurun_out/r06s/full_{i}.json").read().strip().splitlines()[-1]); b=json.loads(open(f"gpurun_out/r06s/bench_{i}.json").read().strip().splitlines()[-1])
    print("--full", round(j["value"],1), "core", round(b["value"],1))
PY
