for t in t1x2 t2x2 t1x4 t2x1; do HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$t.so python profiles/micro/sor_one.py 256 512 rbgs=tile; done
