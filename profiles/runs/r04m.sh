mkdir -p gpurun_out/r04m; rm -f gpurun_out/r04m/*
for a in "256 2 1" "128 2 1" "plume1024 8 1 --partition"; do timeout 300 python3 profiles/micro/dist_overhead.py $a >> gpurun_out/r04m/dist_overhead.jsonl 2>> gpurun_out/r04m/err.log; done
timeout 200 python3 profiles/micro/dist_profile.py rank 1 0 2>/dev/null | grep "ms per"
timeout 200 python3 profiles/micro/dist_profile.py single 2>/dev/null | grep "ms per"
