timeout 900 python3 tests/manual/stress_mirror_processes.py 3 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu" | tail -14
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29552 bench.py --gpus 2 --share-one-gpu --steps 10 --warmup 2 2>&1 | grep "^{\"metric" | python3 -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print(round(j['value'],1), round(j['ms_per_step'],3), j['config']['parallelism'][:170], j['config']['halo'])"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29554 bench.py --gpus 4 --share-one-gpu --config plume1024 --partition --steps 10 --warmup 2 2>&1 | grep "^{\"metric" | python3 -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print(round(j['value'],1), round(j['ms_per_step'],3), j['config']['parallelism'][:170])"
