mkdir -p gpurun_out/r03f; rm -f gpurun_out/r03f/*
for a in "256 2 4" "256 2 2" "128 2 4" "plume1024 8 4 --partition" "plume1024 8 2 --partition"; do timeout 300 python3 profiles/micro/dist_overhead.py $a >> gpurun_out/r03f/dist_overhead.jsonl 2>> gpurun_out/r03f/err.log; done
for w in 0 10 20 40; do timeout 200 python3 profiles/micro/dist_profile.py rank 4 $w 2>> gpurun_out/r03f/err.log | tail -1 >> gpurun_out/r03f/dist_wire_sweep.txt; done
cat gpurun_out/r03f/dist_wire_sweep.txt
