python profiles/micro/sor_one.py 256 512 plume1024
HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_exp128.so python profiles/micro/sor_one.py 256 512 plume1024
bash profiles/micro/pmc_sor.sh e128 512 $PWD/profiles/micro/exp/libhns_exp128.so 2>&1 | grep -E "FETCH|EA0_RDREQ_sum|TCC_MISS|TCC_HIT"
