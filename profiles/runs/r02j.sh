python profiles/micro/sor_one.py 256 512 plume1024
for m in 32 64 96; do HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_exp$m.so python profiles/micro/sor_one.py 256 512 plume1024; done
