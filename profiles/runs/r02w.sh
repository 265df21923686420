timeout 600 python -m pytest tests -m gpu -q -x --durations=5 2>&1 | tail -14
python bench.py --no-cpu-baseline | python -c "import json,sys; j=json.load(sys.stdin); print('256', j['value'], j['roofline']['frac'], {k:round(1e3*v['ms_per_launch'],1) for k,v in j['roofline']['kernels'].items()})"
