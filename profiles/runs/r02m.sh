timeout 600 python -m pytest tests/test_kernel_variants_gpu.py tests/test_kats.py -m gpu -q -x -k "blocked or fused_sor" 2>&1 | tail -3
python profiles/micro/sor_one.py 256 512 plume1024 rbgs=tile
for t in t1x2 t2x2 t1x4; do HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$t.so python profiles/micro/sor_one.py 256 512 plume1024 rbgs=tile; done
