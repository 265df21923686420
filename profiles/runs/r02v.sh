timeout 300 python -m pytest tests/test_resident_gpu.py -m gpu -q -x 2>&1 | tail -12
timeout 120 python profiles/micro/sor_one.py 64 128 plume rbgs=pair
timeout 120 python profiles/micro/sor_one.py 64 128 plume rbgs=resident
