for i in 1 2 3; do HNS_TEST_DUMP=20 timeout 500 python -m pytest tests/test_dist_gpu.py -m gpu -q -x -k "dense32-2-4" --durations=2 2>&1 | tail -30 | cut -c1-200; done
