timeout 300 python3 profiles/micro/advect_ab.py advect_box 0 1 2>&1 | tail -1
timeout 300 python3 profiles/micro/advect_ab.py advect_box 0 1 128 2>&1 | tail -1
timeout 600 python -m pytest tests/test_kernel_variants_gpu.py -x -q -k "advect" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
