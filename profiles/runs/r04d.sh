timeout 600 python3 profiles/micro/advect_lib_ab.py hnanosolver_amd/lib/libhns.so profiles/micro/exp/libhns_ldv4.so 2>&1 | tail -4
