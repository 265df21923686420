mkdir -p gpurun_out/r03q; rm -f gpurun_out/r03q/*
timeout 1200 python -m pytest tests/test_dist_gpu.py -x -q > gpurun_out/r03q/pytest.txt 2>&1; echo rc $? >> gpurun_out/r03q/pytest.txt
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r03q/pytest.txt | tail -30
