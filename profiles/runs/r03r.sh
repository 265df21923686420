mkdir -p gpurun_out/r03r; rm -f gpurun_out/r03r/*
timeout 1200 python -m pytest tests/test_dist_gpu.py -x -q > gpurun_out/r03r/pytest.txt 2>&1; echo rc $? >> gpurun_out/r03r/pytest.txt
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r03r/pytest.txt | tail -5
for a in "256 2 1" "plume1024 8 1 --partition" "128 2 1"; do timeout 300 python3 profiles/micro/dist_overhead.py $a >> gpurun_out/r03r/dist_overhead.jsonl 2>> gpurun_out/r03r/err.log; done
