timeout 300 python -m pytest tests/test_dist_gpu.py -x -q -k "rccl_carries or loopback" 2>&1 | tail -15
