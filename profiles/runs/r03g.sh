mkdir -p gpurun_out/r03g; rm -f gpurun_out/r03g/*
timeout 900 python -m pytest tests/test_dist_gpu.py -x -q 2>&1 | tail -3
for a in "256 2 4" "128 2 4" "plume1024 8 4 --partition"; do timeout 300 python3 profiles/micro/dist_overhead.py $a >> gpurun_out/r03g/dist_overhead.jsonl 2>> gpurun_out/r03g/err.log; done
for w in 0 10 20 40; do timeout 200 python3 profiles/micro/dist_profile.py rank 4 $w 2>> gpurun_out/r03g/err.log | tail -1 >> gpurun_out/r03g/dist_wire_sweep.txt; done
cat gpurun_out/r03g/dist_wire_sweep.txt
