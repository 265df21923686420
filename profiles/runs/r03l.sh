mkdir -p gpurun_out/r03l
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r03l/pytest.txt 2>&1; echo rc $? >> gpurun_out/r03l/pytest.txt
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r03l/pytest.txt | tail -5
