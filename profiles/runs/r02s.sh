python profiles/micro/sor_one.py 64 > /dev/null 2>&1
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
python profiles/micro/sor_one.py 512 
