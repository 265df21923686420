python profiles/micro/dist_profile.py single 2>&1 | tail -1
for w in 0 20; do for k in 4; do python profiles/micro/dist_profile.py rank $k $w 2>&1 | tail -1; done; done
python bench.py --config 64 --no-cpu-baseline | python -c "import json,sys; j=json.load(sys.stdin); print('64', j['value'], j['roofline']['ms_per_launch'])"
