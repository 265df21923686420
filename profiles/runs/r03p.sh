mkdir -p gpurun_out/r03p; rm -f gpurun_out/r03p/*
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29559 bench.py --gpus 4 --share-one-gpu --config plume1024 --partition --steps 10 --warmup 2 > gpurun_out/r03p/bench_share_plume.txt 2>&1; echo rc $? >> gpurun_out/r03p/bench_share_plume.txt
grep "^{\"metric\|^rc" gpurun_out/r03p/bench_share_plume.txt | cut -c1-330
for i in 1 2 3 4 5 6; do
timeout 600 python -m pytest tests/test_dist_gpu.py -x -q -k "one_process_per_rank" > gpurun_out/r03p/pytest$i.txt 2>&1; echo rc $? >> gpurun_out/r03p/pytest$i.txt
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r03p/pytest$i.txt | tail -2 | tr '\n' ' '; echo
done
