mkdir -p gpurun_out/r04u; rm -f gpurun_out/r04u/*
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04u/pytest.txt 2>&1; echo rc $? >> gpurun_out/r04u/pytest.txt
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r04u/pytest.txt | tail -3
for a in "256 2 1" "128 2 1" "plume1024 8 1 --partition"; do timeout 300 python3 profiles/micro/dist_overhead.py $a >> gpurun_out/r04u/dist_overhead.jsonl 2>> gpurun_out/r04u/err.log; done
timeout 600 python bench.py 2>&1 | tail -1 | cut -c1-200
