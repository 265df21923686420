mkdir -p gpurun_out/r03y; rm -f gpurun_out/r03y/*
for i in 1 2 3 4 5; do
timeout 600 python -m pytest tests/test_dist_gpu.py -x -q -k "one_process_per_rank or local_ranks" > gpurun_out/r03y/pytest$i.txt 2>&1; echo rc $? >> gpurun_out/r03y/pytest$i.txt
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r03y/pytest$i.txt | tail -2 | tr '\n' ' '; echo
done
