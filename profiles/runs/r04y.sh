mkdir -p gpurun_out/r04y; rm -f gpurun_out/r04y/*
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04y/pytest.txt 2>&1; echo rc $? >> gpurun_out/r04y/pytest.txt
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r04y/pytest.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
