mkdir -p gpurun_out/r03o; rm -f gpurun_out/r03o/*
for n in 2 4; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2955$n bench.py --gpus $n --share-one-gpu --steps 10 --warmup 2 > gpurun_out/r03o/bench_share$n.txt 2>&1; echo rc $? >> gpurun_out/r03o/bench_share$n.txt
grep "^{\"metric\|^rc" gpurun_out/r03o/bench_share$n.txt | cut -c1-330
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29559 bench.py --gpus 4 --share-one-gpu --config plume1024 --partition --steps 10 --warmup 2 > gpurun_out/r03o/bench_share_plume.txt 2>&1; echo rc $? >> gpurun_out/r03o/bench_share_plume.txt
grep "^{\"metric\|^rc" gpurun_out/r03o/bench_share_plume.txt | cut -c1-330
