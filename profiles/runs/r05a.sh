for pad in 0 3000 6000 10000 16000; do timeout 300 python3 profiles/micro/sor_one.py 512 rbgs=pair sor_lds_pad=$pad 2>/dev/null | tail -1; done
for pad in 0 3000 6000 10000; do timeout 300 python3 profiles/micro/sor_one.py plume1024 sor_lds_pad=$pad 2>/dev/null | tail -1; done
for pad in 0 3000 6000; do timeout 300 python3 profiles/micro/sor_one.py 256 sor_lds_pad=$pad 2>/dev/null | tail -1; done
