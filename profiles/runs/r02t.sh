for i in 1 2; do timeout 900 python -m pytest tests -m gpu -q -x --durations=12 2>&1 | tail -18; done
