for i in 1 2 3 4; do timeout 200 python -m pytest tests/test_dist_gpu.py -m gpu -q -x 2>&1 | tail -2; done
timeout 600 python -m pytest tests -m gpu -q -x --durations=3 2>&1 | tail -8
