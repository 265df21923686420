mkdir -p gpurun_out/r04a; rm -f gpurun_out/r04a/*
timeout 1800 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04a/pytest.txt 2>&1; echo rc $? >> gpurun_out/r04a/pytest.txt
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r04a/pytest.txt | tail -12
