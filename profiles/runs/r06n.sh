# round 6: far taps through the neighbour tables in two hops (up to two leaves away) instead of hash walks: parity tests, then the stage series of the full substep and bench.py --full
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06n; mkdir -p $O
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_kernel_variants_gpu.py tests/test_operators_gpu.py tests/test_ref_kernels_gpu.py tests/test_dist_gpu.py -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 300 python profiles/micro/full_stage_series.py 256 40 1 > $O/series.json 2>$O/series.err
timeout 300 python bench.py --full > $O/full_256.json 2>$O/full.err
python - <<'PY'
import json
a=json.load(open("gpurun_out/r06n/series.json")); b=json.load(open("profiles/r06_full256_stage_series.json"))
k=[x for x in a if x.startswith("us per")][0]
for i,(x,y) in enumerate(zip(a[k],b[k])): print(i, "now", x, "| before", y)
j=json.loads(open("gpurun_out/r06n/full_256.json").read().strip().splitlines()[-1]); print("bench --full:", j["value"], j["ms_per_step"])
PY
