python profiles/micro/sor_one.py 64 > /dev/null 2>&1
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
python profiles/micro/sor_one.py 256 512 plume plume1024
python bench.py --no-cpu-baseline --config plume1024 | python -c "import json,sys; j=json.load(sys.stdin); print('plume1024', j['value'], j['roofline']['frac'])"
python bench.py --no-cpu-baseline --config 512 --steps 5 | python -c "import json,sys; j=json.load(sys.stdin); print('512', j['value'], j['roofline']['frac'])"
