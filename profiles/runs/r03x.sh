mkdir -p gpurun_out/r03x; rm -f gpurun_out/r03x/*
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29554 bench.py --gpus 4 --share-one-gpu --config plume1024 --partition --steps 10 --warmup 2 > gpurun_out/r03x/bench_share_plume.txt 2>&1; echo rc $? >> gpurun_out/r03x/bench_share_plume.txt
grep "^{\"metric\|^rc" gpurun_out/r03x/bench_share_plume.txt | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['value'], j['ms_per_step'], j['config']['parallelism'], j['config']['halo'])
    else: print(l.strip())"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29552 bench.py --gpus 2 --share-one-gpu --steps 10 --warmup 2 > gpurun_out/r03x/bench_share2.txt 2>&1; echo rc $? >> gpurun_out/r03x/bench_share2.txt
grep "^{\"metric\|^rc" gpurun_out/r03x/bench_share2.txt | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['value'], j['ms_per_step'], j['config']['parallelism'][:120])
    else: print(l.strip())"
