for i in 1 2 3; do timeout 800 python profiles/micro/dist_first_use.py 2>&1 | tail -1; done
