timeout 900 python -m pytest tests -m gpu -q -x --durations=4 2>&1 | tail -12
