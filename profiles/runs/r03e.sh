timeout 300 python -m pytest tests/test_dist_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python profiles/micro/dist_overhead.py plume1024 8 4 --partition | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('plume1024/8', 'single', j['single_gpu_substep_ms'], 'lockstep', j['all_ranks_lockstep_ms'], 'lone', j['one_rank_loopback'])"
timeout 300 python profiles/micro/dist_overhead.py 256 2 4 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('256/2', 'single', j['single_gpu_substep_ms'], 'lockstep', j['all_ranks_lockstep_ms'], 'lone', j['one_rank_loopback'])"
