for i in 1 2; do
timeout 600 python -X faulthandler -m pytest tests/test_dist_gpu.py -m gpu -q -x --durations=4 -o faulthandler_timeout=60 2>&1 | tail -40
done
