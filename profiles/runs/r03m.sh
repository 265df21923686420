mkdir -p gpurun_out/r03m
timeout 900 python -m pytest tests/test_dist_gpu.py -x -q -k "one_process_per_rank" > gpurun_out/r03m/pytest.txt 2>&1; echo rc $? >> gpurun_out/r03m/pytest.txt
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r03m/pytest.txt | tail -40
