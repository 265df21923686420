timeout 300 python3 profiles/micro/dist_ab.py alternate 0 1 2>/dev/null | tail -1
timeout 300 python3 profiles/micro/dist_ab.py alternate 0 1 1 128 2>/dev/null | tail -1
timeout 900 python -m pytest tests/test_dist_gpu.py -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -2
