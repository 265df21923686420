timeout 600 python -m pytest tests/test_kernel_variants_gpu.py tests/test_kats.py -m gpu -q -x -k "blocked or fused_sor or wave_per_leaf_pair" 2>&1 | tail -5
python profiles/micro/sor_one.py 128 256 512 plume plume1024
python profiles/micro/sor_one.py 128 256 512 plume plume1024 rbgs=tile
