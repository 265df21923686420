for i in 1 2 3; do timeout 300 python -m pytest tests/test_dist_gpu.py -m gpu -q -x --durations=3 2>&1 | tail -6; done
