for i in 1 2; do
timeout 200 python3 profiles/micro/dist_ab.py alternate 1 1 2>/dev/null | tail -1
HNS_LIBRARY=profiles/micro/exp/libhns_m8.so timeout 200 python3 profiles/micro/dist_ab.py alternate 1 1 2>/dev/null | tail -1
done
HNS_LIBRARY=profiles/micro/exp/libhns_m8.so timeout 200 python3 profiles/micro/dist_ab.py alternate 1 1 1 128 2>/dev/null | tail -1
