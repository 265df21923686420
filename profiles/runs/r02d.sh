mkdir -p gpurun_out/r02d
python -m pytest tests/test_dist_gpu.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r02d/tests.log
tail -5 gpurun_out/r02d/tests.log
for k in 4 2 1; do python profiles/micro/dist_overhead.py 256 2 $k >> gpurun_out/r02d/overhead.jsonl 2>> gpurun_out/r02d/err.log; done
python profiles/micro/dist_overhead.py 128 2 4 >> gpurun_out/r02d/overhead.jsonl 2>> gpurun_out/r02d/err.log
python profiles/micro/dist_overhead.py plume1024 8 4 --partition >> gpurun_out/r02d/overhead.jsonl 2>> gpurun_out/r02d/err.log
python - <<'PY'
import json
for l in open('gpurun_out/r02d/overhead.jsonl'):
    j=json.loads(l); print(j['config'], j['world'], 'k',j['sweeps_per_exchange'], 'single',j['single_gpu_substep_ms'],'lockstep_ovh',j['lockstep_overhead'], j['one_rank_loopback'])
PY
grep -v amdgpu.ids gpurun_out/r02d/err.log | tail -5
