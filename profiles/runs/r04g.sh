mkdir -p gpurun_out/r04g; rm -f gpurun_out/r04g/*
for i in 1 2 3; do timeout 900 python -m pytest tests/test_dist_gpu.py -q -k "one_process_per_rank" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -2 | tr '\n' ' '; echo; done
for t in "auto" "ipc --sweeps-per-exchange 4"; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29552 bench.py --gpus 2 --share-one-gpu --transport $t --steps 10 --warmup 2 > gpurun_out/r04g/b.txt 2>&1; echo rc $? >> gpurun_out/r04g/b.txt
grep "^{\"metric\|^rc" gpurun_out/r04g/b.txt | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(round(j['value'],1), round(j['ms_per_step'],3), j['config']['parallelism'][:150])
    else: print(l.strip())"
done
