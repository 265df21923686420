for i in 1 2 3; do timeout 900 python -m pytest tests/test_dist_gpu.py -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -2 | tr '\n' ' '; echo; done
