echo product; timeout 200 python3 profiles/micro/dist_profile.py rank 1 0 2>/dev/null | grep "ms per"
for v in m5 m7; do echo $v; HNS_LIBRARY=profiles/micro/exp/libhns_$v.so timeout 200 python3 profiles/micro/dist_profile.py rank 1 0 2>/dev/null | grep "ms per"; done
echo product; timeout 200 python3 profiles/micro/dist_profile.py rank 1 0 2>/dev/null | grep "ms per"
