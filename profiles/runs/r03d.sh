timeout 900 python -m pytest tests -m gpu -q -x --durations=3 2>&1 | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py > gpurun_out/r03d_bench.json 2> gpurun_out/r03d_bench.err; python -c "
import json; j=json.load(open('gpurun_out/r03d_bench.json')); print(j['value'], j['roofline']['frac'], j['roofline']['traffic'], j['cpu_baseline']['sample'][:60])"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
