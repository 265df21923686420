for a in "4 0" "1 0"; do timeout 200 python3 profiles/micro/dist_profile.py rank $a 2>/dev/null | grep "ms per substep"; done
timeout 200 python3 profiles/micro/dist_profile.py single 2>/dev/null | grep "ms per substep"
timeout 300 python3 profiles/micro/dist_overhead.py 256 2 1 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('lone', j['one_rank_loopback'], 'single', j['single_gpu_substep_ms'])"
timeout 300 python3 profiles/micro/dist_overhead.py plume1024 8 1 --partition 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('lone', j['one_rank_loopback'], 'single', j['single_gpu_substep_ms'])"
timeout 300 python3 profiles/micro/dist_overhead.py 128 2 1 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('lone', j['one_rank_loopback'], 'single', j['single_gpu_substep_ms'])"
