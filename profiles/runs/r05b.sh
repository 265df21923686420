for pad in 0 1500 3400 6000; do timeout 300 python3 profiles/micro/sor_one.py 512 sor_lds_pad=$pad 2>/dev/null | tail -1; done
