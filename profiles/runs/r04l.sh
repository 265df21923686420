for i in 1 2; do timeout 900 python -m pytest tests/test_dist_gpu.py -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -2 | tr '\n' ' '; echo; done
timeout 200 python3 profiles/micro/dist_ab.py alternate 1 1 2>/dev/null | tail -1
timeout 200 python3 profiles/micro/dist_ab.py alternate 1 1 1 128 2>/dev/null | tail -1
timeout 900 python3 tests/manual/stress_mirror_processes.py 2 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu" | tail -9
