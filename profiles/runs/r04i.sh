timeout 1500 python3 tests/manual/stress_mirror_processes.py 6 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu" | tail -30
